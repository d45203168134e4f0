#!/bin/bash
# round 4 (second session): the tuner's clock warm-up (five steady groups) against round 3's
# (two groups agreeing): the driver's window, alternating processes, same box
set -u
out=gpurun_out/r04_warm; mkdir -p $out
f="--no-cpu-baseline --no-traffic --no-substrate-api --steps 20 --warmup 5"
for i in 1 2 3 4; do
  MP_BENCH_ALLOW_DEV_ENV=1 MP_ENGINE_LIB=$PWD/meltingpot_amd/lib/libmp_engine_prev.so timeout 200 python bench.py $f > $out/prev_$i.json 2>/dev/null
  timeout 200 python bench.py $f > $out/new_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r04_warm/*.json")):
  try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    pl = d.get("placement") or {}
    print(p.split("/")[-1], round(d["ms_per_step"] * 1e3, 1), round(d["roofline"]["frac"], 3), "picked dry", pl.get("dry_launch_us")[pl.get("picked")], "feeders", d["plan"].get("feeders"), "B", d["plan"]["batch_worlds"])
  except Exception as ex:
    print(p, "unreadable", ex)
PY
