#!/bin/bash
# round 5, call 5: is a view's speed a matter of how SCATTERED its physical chunks are?  The study on a
# fresh box, then again behind the full-size GPU tests (the state BENCH_r04's driver box was in)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call5; mkdir -p $O
cd $R
( time timeout 600 python tools/alloc_method_study.py --processes 6 --configs clean_up_both,commons_agents --out $O/fresh ) > $O/alloc_fresh.log 2>&1
echo "alloc fresh rc=$?"; tail -24 $O/alloc_fresh.log
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_matrix.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
( time timeout 600 python tools/alloc_method_study.py --processes 6 --configs clean_up_both,commons_agents --out $O/after_suite ) > $O/alloc_after.log 2>&1
echo "alloc after rc=$?"; tail -24 $O/alloc_after.log
