#!/bin/bash
# round 5, call 13: gift_refinements (a seventh Lua level) against the oracle; the reference's
# per-substrate conformance test on the HIP engine for all 25 substrates; the next-orders tests
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call13; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_gift.py tests/test_every_substrate.py tests/test_gpu_next_orders.py -m gpu -x -q --durations=6 ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -40 $O/pytest.log
