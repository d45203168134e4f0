#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03h; mkdir -p $O
bash tools/gpu_tests.sh 300 900 2>&1 | tee $O/tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_ab.sh "- prev" "" 2
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_timeline.so timeout 120 python tools/gpu_timeline.py clean_up 4096 world > $O/timeline.txt 2>&1; grep -A13 "slot 0" $O/timeline.txt | cut -c1-160
