#!/bin/bash
# round 6, call 5: every pack at 2048 worlds x 900 steps against the oracle (sampled worlds replayed from hashed
# actions: tests/tools/deep_soak.py); the GPU suite on the round's last library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call5; mkdir -p $O
( time timeout 2400 python tests/tools/deep_soak.py 2048 900 ) > $O/deep_soak.txt 2>&1; echo "deep soak rc=$?"; grep -v amdgpu.ids $O/deep_soak.txt | tail -40
