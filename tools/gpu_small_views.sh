#!/bin/bash
# dev helper: two-player substrates (views under 64 KB a world): two launches vs the fused launch with many feeders
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
fmt='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; print(sys.argv[1], "ms/step %.4f" % d["ms_per_step"], {a: round(b*1e3,1) for a,b in k.items() if a in ("frame","step","render")}, "%.0fM" % (d["value"]/1e6))'
run() { timeout -k 5 90 python -u bench.py $1 --no-cpu-baseline --no-traffic --steps 100 $2 2>/dev/null | tail -1 | python -c "$fmt" "[$1] ${2:12:40}"; }
for cfg in "--substrate prisoners_dilemma_in_the_matrix__repeated --obs agents --worlds 16384" "--substrate coins --obs agents --worlds 16384" "--substrate running_with_scissors_in_the_matrix__repeated --obs agents --worlds 16384"; do
  run "--unfused" "$cfg"
  for plan in ${PLANS:-"batch_worlds=8,feeders=8,waves=16" "batch_worlds=8,feeders=4,waves=16" "batch_worlds=8,feeders=8,waves=12" "batch_worlds=8,feeders=8,waves=14" "batch_worlds=8,feeders=8,waves=16,late_feeder_prio=1" "batch_worlds=8,feeders=8,waves=16,late_feeder_prio=4" "batch_worlds=8,feeders=8,waves=16"}; do
    run "--fused --dev-plan $plan" "$cfg"
  done
done
