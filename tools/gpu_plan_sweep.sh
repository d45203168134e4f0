#!/bin/bash
# dev helper: fused frame time over (batch B, feeders F, waves) on one config; sorted.
# usage: tools/gpu_plan_sweep.sh "<bench args>" "<B:F list>" "<waves list>"
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1 MP_BENCH_ALLOW_DEV_ENV=1
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f" % (d["kernels_ms"]["frame"]*1e3), sys.argv[1])'
for bf in $2; do for wv in $3; do
  timeout -k 5 40 python -u bench.py --dev-plan batch_worlds=${bf%:*},feeders=${bf#*:},waves=$wv --no-cpu-baseline --no-traffic --steps 60 --warmup 10 --fused $1 2>/dev/null | tail -1 | python -c "$fmt" "B:F=$bf waves=$wv"
done; done | sort -n
