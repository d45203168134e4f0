#!/bin/bash
# round 5, call 1: the GPU suite (new: ring, at-size plans, error paths, scenario replay) and a bench line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call1; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -40 $O/pytest.log
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 6000 $O/bench.json; tail -5 $O/bench.err
