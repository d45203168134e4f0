#!/bin/bash
# round 4: the headline bench with the tuner on / the stock plan forced / the single-world ring forced
set -u
out=gpurun_out/r04_bench_plans; mkdir -p $out
f="--no-cpu-baseline --no-traffic --no-substrate-api"
for i in 1 2; do
  python bench.py $f > $out/tuned_$i.json 2>/dev/null
  python bench.py $f --dev-plan static_pct=100 > $out/stock_$i.json 2>/dev/null
  python bench.py $f --dev-plan batch_worlds=1,ring_batches=8 > $out/ring_$i.json 2>/dev/null
  python bench.py $f --place 1 > $out/first_alloc_$i.json 2>/dev/null
  python bench.py $f --steps 20 --warmup 5 > $out/driver_flags_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r04_bench_plans/*.json")):
  try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    pl = d.get("placement") or {}
    print(p.split("/")[-1], round(d["ms_per_step"] * 1e3, 1), round(d["roofline"]["frac"], 3), "dry", pl.get("dry_launch_us"), pl.get("picked"))
  except Exception as ex:
    print(p, "unreadable", ex)
PY
