#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03j; mkdir -p $O
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "frame %.1f us" % (d["kernels_ms"]["frame"]*1e3), "frac %.3f" % d["roofline"]["frac"])'
run() { timeout -k 5 90 python -u bench.py --dev-plan $1 --no-cpu-baseline --no-traffic --steps 100 $2 2>/dev/null | tail -1 | python -c "$fmt" "[$1] ${2:12:30}"; }
C="--substrate commons_harvest__open --obs agents"
timeout 60 python -u bench.py --dev-plan verbose=1 --no-cpu-baseline --no-traffic --steps 20 $C 2>&1 | grep "stepping + drawing, agents"
for i in 1 2 3; do run verbose=0 "$C"; done
for bf in 3:3 3:2 3:6 3:3 3:2; do run batch_worlds=${bf%:*},feeders=${bf#*:},waves=16 "$C"; done
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_timeline.so timeout 120 python tools/gpu_timeline.py commons_harvest__open 4096 agents > $O/timeline_commons.txt 2>&1; grep -A17 "slot 0" $O/timeline_commons.txt | cut -c1-260
