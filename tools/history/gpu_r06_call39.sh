#!/bin/bash
# round 6, call 39: the whole GPU suite and smoke on the round's last commit (MpInfo.plan_late_priority came after call 34)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call39; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -x -q --durations=5 ) > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -4 $O/pytest_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-160
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench rc=$?"
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r06_call39/bench_driver_flags.json").read().strip().splitlines()[-1])
print("headline", round(l["value"] / 1e6, 1), "M", round(l["roofline"]["avg_launch_ms"] * 1e3, 1), "us", round(l["roofline"]["frac"], 3), l["plan"], l["placement"]["dry_launch_us"])
print({k: (round(v["avg_launch_ms"] * 1e3, 1), round(v["frac"], 3)) for k, v in l["configs"].items()}, round(l["substrate_api"]["avg_launch_ms"] * 1e3, 1), round(l["substrate_api"]["frac"], 3))
PY
