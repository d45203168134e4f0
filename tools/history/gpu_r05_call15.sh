#!/bin/bash
# round 5, call 15: collaborative_cooking (an eighth Lua level, seven substrates) against the
# oracle; the per-substrate conformance test and the next-orders test with the new names
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call15; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests/test_gpu_cook.py tests/test_every_substrate.py tests/test_gpu_next_orders.py -m gpu -x -q --durations=6 ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -40 $O/pytest.log
