#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03m; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_matrix.py -m gpu -q -x --timeout=300 2>&1 | tail -2
bash tools/gpu_ab.sh "- prev" "" 3
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_timeline.so timeout 120 python tools/gpu_timeline.py clean_up 4096 world > $O/timeline.txt 2>&1; grep -A13 "slot 0" $O/timeline.txt | grep "wave  0\|wave  4\|wave  8\|wave  9\|wave 10\|wave 11" | cut -c1-150
