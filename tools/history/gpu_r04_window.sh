#!/bin/bash
# round 4 (second session): what does the driver's window (20 steps after 5) see that 200 after 100 does not?
set -u
out=gpurun_out/r04_window; mkdir -p $out
f="--no-cpu-baseline --no-traffic --no-substrate-api"
for i in 1 2 3; do
  for w in "20 5" "20 100" "200 5" "200 100"; do
    set -- $w
    timeout 200 python bench.py $f --steps $1 --warmup $2 > $out/s$1_w$2_$i.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r04_window/*.json")):
  try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    pl = d.get("placement") or {}
    print(p.split("/")[-1], "wall", round(d["ms_per_step"] * 1e3, 1), "events", round(d["kernels_ms"]["frame"] * 1e3, 1), "picked probe", pl["dry_launch_us"][pl["picked"]], "cands", pl["candidates"])
  except Exception as ex:
    print(p, "unreadable", ex)
PY
