#!/bin/bash
# round 6, call 28: every pack at 2048 worlds x 900 steps against the oracle on the final library (tests/tools/deep_soak.py:
# sampled worlds replayed from hashed actions; the bound view after every third of the run), then the long soak of all packs
# with auto-reset (tests/tools/soak.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call28; mkdir -p $O
( time timeout 2400 python tests/tools/deep_soak.py 2048 900 ) > $O/deep_soak.txt 2>&1; echo "deep soak rc=$?"; grep -v amdgpu.ids $O/deep_soak.txt | tail -4 | cut -c1-300
( time timeout 2400 python tests/tools/soak.py ) > $O/soak.txt 2>&1; echo "soak rc=$?"; grep -v amdgpu.ids $O/soak.txt | tail -5 | cut -c1-300
