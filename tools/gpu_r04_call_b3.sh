#!/bin/bash
# round 4 (second session), last call: smoke(), the GPU suite, the round's profile set
set -u
out=gpurun_out/r04_b3; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $out/smoke.log | grep -v amdgpu
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -3 $out/pytest_gpu.log
bash tools/profile_round.sh r04c > $out/profile_round.log 2>&1; echo "profile rc $?"
timeout 300 python bench.py --steps 20 --warmup 5 > $out/driver_flags.json 2> $out/driver_flags.err; echo "bench rc $?"
python - <<'PY'
import json
for p in ("gpurun_out/prof_r04c/clean_up_world.bench.json", "gpurun_out/prof_r04c/commons_agents.bench.json", "gpurun_out/prof_r04c/territory_agents.bench.json", "gpurun_out/r04_b3/driver_flags.json"):
  try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print(p.split("/")[-1], round(d["value"] / 1e6, 1), round(d["ms_per_step"] * 1e3, 1), round(d["roofline"]["frac"], 3), d["roofline"]["traffic"], d.get("plan"))
    if "substrate_api" in d: print("  api", round(d["substrate_api"]["ms_per_step"]*1e3,1), round(d["substrate_api"]["frac"],3), d["substrate_api"]["plan"])
  except Exception as ex:
    print(p, "unreadable", ex)
PY
