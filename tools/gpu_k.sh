#!/bin/bash
# dev helper: renderer-count sensitivity.  $1 = bench args; runs draw-only at 12 waves and fused at several B:F
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1 MP_BENCH_ALLOW_DEV_ENV=1
fmt='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; print(sys.argv[1], "frame %.1f" % (k["frame"]*1e3), ("step %.1f render %.1f" % (k["step"]*1e3, k["render"]*1e3)) if "step" in k else "")'
CFG="$1"
run() { timeout -k 5 60 python -u bench.py --no-cpu-baseline --no-traffic --steps 60 $CFG $2 2>/dev/null | tail -1 | python -c "$fmt" "$1"; }
for wf in 16:4 16:2 12:4 12:2 10:2 8:2; do
  MP_RENDER_WAVES=${wf%:*} MP_RENDER_FEEDERS=${wf#*:} run "unfused waves:F=$wf" --unfused
done
for geo in 4:4 3:3 3:2 6:4 6:3 2:4; do
  MP_RENDER_WPB=${geo%:*} MP_RENDER_FEEDERS=${geo#*:} run "fused B:F=$geo"
done
