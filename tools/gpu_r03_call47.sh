#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_surface.py -m gpu -q -x --timeout=240 --timeout-method=thread -k "placed or layer or raw" 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1800 $O/bench_default.json
