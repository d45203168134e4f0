#!/bin/bash
# round 5, call 29: the bench lines of the three levels whose stock per-agent plan became 4 : 4, and the tests that force / assert plans
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=.; O=gpurun_out/r05_four; mkdir -p $O
for s in coop_mining gift_refinements externality_mushrooms__dense; do
  for k in 1 2; do
    timeout 300 python bench.py --substrate $s --obs agents --no-cpu-baseline --no-traffic > $O/${s}_$k.json 2>> $O/err.log
    python - <<PY
import json
l = json.loads(open("$O/${s}_$k.json").read().strip().splitlines()[-1])
print("$s", $k, round(l["value"] / 1e6, 1), round(l["ms_per_step"] * 1000, 1), round(l["roofline"]["frac"], 3), l["plan"], sorted(l["placement"]["dry_launch_us"])[:2])
PY
  done
done
( time timeout 1200 python -m pytest tests -q -m gpu -x -k "coop or gift or mushroom or geometry or plan" --tb=short ) > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
