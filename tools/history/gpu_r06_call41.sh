#!/bin/bash
# round 6, call 41: the deep soak at twice the batch and a longer horizon on the final library: every pack, 4096 worlds x 1500
# steps, the tuned plan (team order, pause, priority as mp_tune picks them), sampled worlds replayed by the oracle
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call41; mkdir -p $O
( time timeout 3000 python tests/tools/deep_soak.py 4096 1500 ) > $O/deep_soak_4096x1500.txt 2>&1; echo "deep soak rc=$?"; grep -v amdgpu.ids $O/deep_soak_4096x1500.txt | grep -E "deep soak|DIFFER" | head; grep -c "replayed: ok" $O/deep_soak_4096x1500.txt
