#!/bin/bash
# round 3, first GPU call: whole -m gpu suite, smoke, the three bench lines,
# store-ceiling sweep, A/B of per-slot vs whole-batch waits
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03a; mkdir -p $O
bash tools/gpu_tests.sh 300 900 2>&1 | tee $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout 300 python bench.py > $O/bench_clean_up.json 2> $O/bench_clean_up.err; echo "bench rc=$?"; tail -c 1500 $O/bench_clean_up.json
timeout 120 python bench.py --no-cpu-baseline --no-traffic --substrate commons_harvest__open --obs agents > $O/bench_commons.json 2>/dev/null; tail -c 600 $O/bench_commons.json
timeout 120 python bench.py --no-cpu-baseline --no-traffic --substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5 > $O/bench_territory.json 2>/dev/null; tail -c 600 $O/bench_territory.json
timeout 200 tools/ubench/store_ceiling > $O/store_ceiling.md 2>&1; echo "ubench rc=$?"; cat $O/store_ceiling.md
bash tools/gpu_ab.sh "- batchwait" "" 2 2>&1 | tee $O/ab_slotwait.txt
