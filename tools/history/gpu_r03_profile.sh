#!/bin/bash
# round 3 profile set: tests + smoke, bench lines, kernel traces + HBM traffic, timeline, SQ counters
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03p; mkdir -p $O
bash tools/gpu_tests.sh 300 900 2>&1 | tee $O/tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python -m pytest tests/test_gpu_surface.py -m gpu -q -s -k zap_storm 2>&1 | grep "fullest"
bash tools/profile_round.sh r03
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_timeline.so timeout 120 python tools/gpu_timeline.py clean_up 4096 world > $O/timeline.txt 2>&1; grep "workgroup slot" $O/timeline.txt
bash tools/pmc_sq_r02.sh clean_up_world "" 2>&1 | tail -4
timeout 120 python bench.py --no-cpu-baseline --no-traffic --substrate prisoners_dilemma_in_the_matrix__arena --obs agents --worlds 8192 2>/dev/null | tail -1 > $O/bench_pd_arena.json; tail -c 500 $O/bench_pd_arena.json
