#!/bin/bash
# dev helper: draw-only geometry (waves:loaders) and fused B:F on one config, same box
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1 MP_BENCH_ALLOW_DEV_ENV=1
fmt='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; print(sys.argv[1], "frame %.1f" % (k["frame"]*1e3), ("step %.1f render %.1f" % (k["step"]*1e3, k["render"]*1e3)) if "step" in k else "")'
CFG="$1"
run() { timeout -k 5 60 python -u bench.py --dev-plan waves=${wf%:*},feeders=${wf#*:} --no-cpu-baseline --no-traffic --steps 80 $CFG $2 2>/dev/null | tail -1 | python -c "$fmt" "$1"; }
for wf in $2; do
  run "draw-only waves:F=$wf" --unfused
done
for wf in $3; do
  run "fused waves:F=$wf" --fused
done
