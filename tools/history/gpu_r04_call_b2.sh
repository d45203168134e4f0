#!/bin/bash
# round 4 (second session): the GPU suite on the DMA head / host-computed launch constants,
# then the bench lines with the driver's flags and the defaults
set -u
out=gpurun_out/r04_b2; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -4 $out/pytest_gpu.log
timeout 400 python bench.py --steps 20 --warmup 5 > $out/driver_flags.json 2> $out/driver_flags.err; echo "bench rc $?"
timeout 400 python bench.py --no-cpu-baseline --no-traffic > $out/default.json 2> $out/default.err; echo "bench rc $?"
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r04_b2/*.json")):
  try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    pl = d.get("placement") or {}
    print(p.split("/")[-1], round(d["ms_per_step"] * 1e3, 1), round(d["roofline"]["frac"], 3), d.get("plan"), "dry", pl.get("dry_launch_us"), pl.get("picked"))
    if "substrate_api" in d: print("  api", round(d["substrate_api"]["ms_per_step"]*1e3,1), round(d["substrate_api"]["frac"],3), d["substrate_api"]["plan"])
  except Exception as ex:
    print(p, "unreadable", ex)
PY
