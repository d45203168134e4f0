#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03f; mkdir -p $O
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "frame %.1f us" % (d["kernels_ms"]["frame"]*1e3), "frac %.3f" % d["roofline"]["frac"])'
run() { timeout -k 5 90 python -u bench.py --dev-plan $1 --no-cpu-baseline --no-traffic --steps 100 $2 2>/dev/null | tail -1 | python -c "$fmt" "[$1] ${2:12:30}"; }
T="--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5"
C="--substrate commons_harvest__open --obs agents"
M="--substrate prisoners_dilemma_in_the_matrix__arena --obs agents --worlds 8192 --fused"
for lp in 1 2 3 4; do run late_feeder_prio=$lp "$T"; run late_feeder_prio=$lp "$M"; run late_feeder_prio=$lp "$C"; run late_feeder_prio=$lp ""; done
echo "== clean_up plan sweep"
for bf in 4:4 4:2 4:8 2:4 3:3 3:6 5:5; do for wv in 10 12 14 16; do run batch_worlds=${bf%:*},feeders=${bf#*:},waves=$wv ""; done; done | sort -t' ' -k4 -n | head -12
echo "== commons plan sweep"
for bf in 3:6 3:3 4:4 4:8 2:4; do for wv in 14 16; do run batch_worlds=${bf%:*},feeders=${bf#*:},waves=$wv "$C"; done; done | sort -t' ' -k6 -n | head -6
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_timeline.so timeout 120 python tools/gpu_timeline.py clean_up 4096 world > $O/timeline.txt 2>&1; head -20 $O/timeline.txt | cut -c1-900
