#!/bin/bash
# dev helper: frame-kernel geometry sweep (batch B, feeders F) on one config, fused and unfused
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1 MP_BENCH_ALLOW_DEV_ENV=1
fmt='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; print(sys.argv[1], "frame %.1f" % (k["frame"]*1e3), ("step %.1f render %.1f" % (k["step"]*1e3, k["render"]*1e3)) if "step" in k else "")'
CFG="$1"
for geo in $2; do
  for mode in "" "--unfused"; do
    timeout -k 5 60 python -u bench.py --dev-plan batch_worlds=${geo%:*},feeders=${geo#*:} --no-cpu-baseline --no-traffic --steps 60 $CFG $mode 2>/dev/null | tail -1 | python -c "$fmt" "B:F=$geo ${mode:-fused}"
  done
done
