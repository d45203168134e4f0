#!/bin/bash
# round 6, call 18: the new tests (every count of visible planes — the packs rebuilt around the avatars' plane —, paced
# renderers, the paced plan forced at full size) on the library without the old resolve's code path
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call18; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "visible_planes or paced or tuner_plans" --durations=5 ) > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -12 $O/pytest_new.log
