#!/bin/bash
# round 5, call 37: the bench line with the box's own report (rocm-smi) beside it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=.; O=gpurun_out/r05_box; mkdir -p $O
rocm-smi --showcomputepartition --showmemorypartition --showclocks --showpower --showperflevel --json > $O/rocm_smi.json 2> $O/rocm_smi.err; head -c 1500 $O/rocm_smi.json; echo
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?"
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r05_box/bench.json").read().strip().splitlines()[-1])
print(round(l["value"] / 1e6, 1), round(l["ms_per_step"] * 1000, 1), l.get("box"))
PY
