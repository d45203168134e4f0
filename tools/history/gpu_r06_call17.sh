#!/bin/bash
# round 6, call 17: the new tests again (call 16: a pack whose avatars have no sprite is refused — the zero-plane case is
# out); what a headline launch issues, by kind of instruction, old resolve against new (SQ counters)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call17; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "visible_planes or paced or tuner_plans" --durations=5 ) > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -12 $O/pytest_new.log
bash tools/pmc_insts.sh gpurun_out/r06_call17 - v1 2>&1 | grep -v amdgpu.ids | tee $O/pmc_insts.txt | tail -30
