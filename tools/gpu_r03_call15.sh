#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03l; mkdir -p $O
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_timeline.so timeout 120 python tools/gpu_timeline.py clean_up 4096 world > $O/timeline.txt 2>&1; grep -A13 "slot 0" $O/timeline.txt | grep "wave  8\|wave  9\|wave 10\|wave 11" | cut -c1-220
