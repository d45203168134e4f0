#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=300 2>&1 | tail -2
bash tools/gpu_ab.sh "- prev" "" 2
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_timeline.so timeout 120 python tools/gpu_timeline.py clean_up 4096 world > $O/timeline.txt 2>&1; grep -A13 "slot 0" $O/timeline.txt | cut -c1-130
