#!/bin/bash
# dev helper: same-box A/B of engine builds on ONE config.  usage: tools/gpu_ab1.sh "<tags>" "<bench args>" [reps]
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1 MP_BENCH_ALLOW_DEV_ENV=1
fmt='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; print(sys.argv[1], "frame %.1f" % (k["frame"]*1e3), ("step %.1f render %.1f" % (k["step"]*1e3, k["render"]*1e3)) if "step" in k else "")'
for rep in $(seq 1 ${3:-3}); do
  for tag in $1; do
    lib=""; [ "$tag" != "-" ] && lib=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_$tag.so
    MP_ENGINE_LIB=$lib timeout -k 5 60 python -u bench.py --no-cpu-baseline --no-traffic --steps 100 $2 2>/dev/null | tail -1 | python -c "$fmt" "[$tag]"
  done
done
