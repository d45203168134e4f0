#!/bin/bash
# round 4 (second session): the GPU suite at HEAD, then the driver's flags, three processes
# (the call first compared them with 30 / 120 ms of draw-only launches in front of the warm-up
# steps, a bench.py flag that was not kept: 112.4 vs 112.9 us, no difference — the tuner's
# own warm-up is what mattered, tools/gpu_r04_warm.sh)
set -u
out=gpurun_out/r04_b1; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -4 $out/pytest_gpu.log
f="--no-cpu-baseline --no-traffic --no-substrate-api --steps 20 --warmup 5"
for i in 1 2 3; do
  timeout 200 python bench.py $f > $out/run_$i.json 2>/dev/null
done
timeout 400 python bench.py --steps 20 --warmup 5 > $out/driver_full.json 2> $out/driver_full.err; echo "full rc $?"
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r04_b1/*.json")):
  try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    pl = d.get("placement") or {}
    print(p.split("/")[-1], round(d["ms_per_step"] * 1e3, 1), round(d["roofline"]["frac"], 3), d.get("clock_warm"), d.get("plan"), "dry", pl.get("dry_launch_us"), pl.get("picked"))
    if "substrate_api" in d: print("  api", d["substrate_api"]["ms_per_step"], d["substrate_api"]["frac"])
  except Exception as ex:
    print(p, "unreadable", ex)
PY
