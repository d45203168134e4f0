#!/bin/bash
# round 4: the whole GPU suite, then the bench lines (headline placed / first allocation)
set -u
out=gpurun_out/r04_suite; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -15 $out/pytest_gpu.log
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
timeout 600 python bench.py --place 1 --placements 6 --no-cpu-baseline --no-traffic --no-substrate-api > $out/bench_place1.json 2> $out/bench_place1.err; echo "bench rc $?"
python - <<'PY'
import json
for f in ("bench_default", "bench_place1"):
  try:
    d = json.loads(open(f"gpurun_out/r04_suite/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"], d.get("placement"), d["kernels_ms"])
    if "substrate_api" in d: print(" substrate_api", d["substrate_api"])
  except Exception as ex: print(f, "unreadable", ex)
PY
