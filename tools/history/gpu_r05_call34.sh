#!/bin/bash
# round 5, call 34: the new stock plan of WORLD.RGB alone at 16 - 64 KB a world: bench lines and the tests of those levels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=.; O=gpurun_out/r05_world; mkdir -p $O
for s in collaborative_cooking__crowded collaborative_cooking__figure_eight externality_mushrooms__dense; do
  timeout 300 python bench.py --substrate $s --obs world --no-cpu-baseline --no-traffic > $O/${s}.json 2>> $O/err.log
  python - <<PY
import json
l = json.loads(open("$O/${s}.json").read().strip().splitlines()[-1])
print("$s", round(l["value"] / 1e6, 1), round(l["ms_per_step"] * 1000, 1), round(l["roofline"]["frac"], 3), l["plan"])
PY
done
( time timeout 1200 python -m pytest tests -q -m gpu -x -k "cook or mushroom or geometry or plan" --tb=short ) > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
