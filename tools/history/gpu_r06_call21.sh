#!/bin/bash
# round 6, call 21: team dealing among mp_tune's candidates, the test-free copy road, host reciprocals: the whole GPU
# suite, smoke, and the bench line (driver's flags) twice
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call21; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -12 $O/pytest_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-160
for rep in 1 2; do
  ( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "bench $rep rc=$?"; tail -3 $O/bench_$rep.err
done
python - <<'PY'
import json
for f in ("bench_1", "bench_2"):
  l = json.loads(open(f"gpurun_out/r06_call21/{f}.json").read().strip().splitlines()[-1])
  print(f, "headline", round(l["value"] / 1e6, 1), "M", round(l["roofline"]["avg_launch_ms"] * 1e3, 1), "us", round(l["roofline"]["frac"], 3), "traffic", l["roofline"]["traffic"], "plan", l["plan"])
  print("  placement", l["placement"])
  print("  box_fill", json.dumps(l.get("box_fill")))
  sa = l.get("substrate_api") or {}
  print("  substrate_api", round(sa.get("avg_launch_ms", 0) * 1e3, 1), "us", round(sa.get("frac", 0), 3), sa.get("plan"))
  ra = l.get("rollout_api") or {}
  print("  rollout single", round(ra["single"]["events_ms_per_step"] * 1e3, 1), "us", ra["single"]["plan"], "ring", round(ra["ring"]["events_ms_per_step"] * 1e3, 1))
  for k, v in (l.get("configs") or {}).items():
    print("  ", k, round(v["value"] / 1e6, 1), "M", round(v["avg_launch_ms"] * 1e3, 1), "us", round(v["frac"], 3), "of box fill", round(v["box_fill"]["frac_of_box_fill"], 3), v.get("plan"), v["placement"].get("kind"), v["placement"].get("dry_launch_us"))
  print("  cpu_baseline", l.get("cpu_baseline"))
PY
