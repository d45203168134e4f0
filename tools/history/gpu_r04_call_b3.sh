#!/bin/bash
# round 4 (second session), last call: the round's profile set on the fresh box, then smoke()
# and the GPU suite
set -u
out=gpurun_out/r04_b3; mkdir -p $out
bash tools/profile_round.sh ${1:-r04d} > $out/profile_round.log 2>&1; echo "profile rc $?"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $out/smoke.log
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -2 $out/pytest_gpu.log
python - <<'PY'
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04d"
for n in ("clean_up_world", "commons_agents", "territory_agents"):
  p = f"gpurun_out/prof_{tag}/{n}.bench.json"
  try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print(n, round(d["value"] / 1e6, 1), round(d["ms_per_step"] * 1e3, 1), round(d["roofline"]["frac"], 3), d["roofline"]["traffic"], d.get("plan"))
    if "substrate_api" in d: print("  api", round(d["substrate_api"]["ms_per_step"]*1e3,1), round(d["substrate_api"]["frac"],3), d["substrate_api"]["plan"])
  except Exception as ex:
    print(p, "unreadable", ex)
PY
