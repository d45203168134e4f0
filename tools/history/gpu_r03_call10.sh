#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_matrix.py -m gpu -q -x --timeout=300 2>&1 | tail -3
bash tools/gpu_ab.sh "- prev" "" 2
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "frame %.1f us" % (d["kernels_ms"]["frame"]*1e3), "frac %.3f" % d["roofline"]["frac"])'
run() { timeout -k 5 90 python -u bench.py --dev-plan $1 --no-cpu-baseline --no-traffic --steps 100 $2 2>/dev/null | tail -1 | python -c "$fmt" "[$1] ${2:12:30}"; }
T="--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5"
C="--substrate commons_harvest__open --obs agents"
M="--substrate prisoners_dilemma_in_the_matrix__arena --obs agents --worlds 8192 --fused"
echo "== commons plan sweep"
for bf in 3:3 3:2 4:4 4:2 5:5 2:2 3:6; do run batch_worlds=${bf%:*},feeders=${bf#*:},waves=16 "$C"; done
echo "== territory plan sweep"
for bf in 3:3 3:2 4:4 2:4 3:6; do for lp in 2 4; do run batch_worlds=${bf%:*},feeders=${bf#*:},waves=16,late_feeder_prio=$lp "$T"; done; done
echo "== matrix plan sweep"
for bf in 3:6 3:3 4:4 4:8; do for lp in 1 4; do run batch_worlds=${bf%:*},feeders=${bf#*:},waves=16,late_feeder_prio=$lp "$M"; done; done
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_timeline.so timeout 120 python tools/gpu_timeline.py clean_up 4096 world > $O/timeline.txt 2>&1; grep -A13 "slot 0" $O/timeline.txt | cut -c1-200
