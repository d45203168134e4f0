#!/bin/bash
# round 5: which boxes are slow, and what do they say about themselves?  (territory's per-agent launch is the most sensitive)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=.; O=gpurun_out/r05_box; mkdir -p $O
timeout 200 python bench.py --substrate territory__rooms --obs agents --players 9 --worlds 8192 --beam-skew 0.5 --steps 100 --warmup 100 --no-cpu-baseline --no-traffic --no-steady-state > $O/t.json 2>> $O/err.log
python - <<'PY'
import json, socket
l = json.loads(open("gpurun_out/r05_box/t.json").read().strip().splitlines()[-1])
print(socket.gethostname(), round(l["ms_per_step"] * 1000, 1), "dry", min(l["placement"]["dry_launch_us"]), max(l["placement"]["dry_launch_us"]), {k: v for k, v in (l.get("box") or {}).items() if "level" not in k})
PY
rocm-smi --showtemp --showpower --showmaxpower --json 2>/dev/null | head -c 600; echo
