#!/bin/bash
# round 6, call 34: the final library: the whole GPU suite, smoke, the bench line (driver's flags, then the defaults)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call34; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -4 $O/pytest_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-160
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench (driver flags) rc=$?"; tail -3 $O/bench_driver_flags.err | grep real
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err | grep real
python - <<'PY'
import json
for f in ("bench_driver_flags", "bench"):
  l = json.loads(open(f"gpurun_out/r06_call34/{f}.json").read().strip().splitlines()[-1])
  p = lambda d: {k: v for k, v in d.items() if k in ("batch_worlds", "feeders", "pace", "xcd_teams", "late_feeder_priority", "sc1_stores", "pooled_batches")}
  print(f, "headline", round(l["value"] / 1e6, 1), "M", round(l["roofline"]["avg_launch_ms"] * 1e3, 1), "us", round(l["roofline"]["frac"], 3), "traffic", l["roofline"]["traffic"], "plan", p(l["plan"]))
  print("  placement", l["placement"]["dry_launch_us"], l["placement"]["setup_s"], "box_fill", l["box_fill"]["views"], round(l["box_fill"]["frac_of_box_fill"], 3))
  sa = l.get("substrate_api") or {}
  print("  substrate_api", round(sa.get("avg_launch_ms", 0) * 1e3, 1), "us", round(sa.get("frac", 0), 3), p(sa.get("plan")))
  ra = l.get("rollout_api") or {}
  print("  rollout single", round(ra["single"]["events_ms_per_step"] * 1e3, 1), "us", p(ra["single"]["plan"]), "ring", round(ra["ring"]["events_ms_per_step"] * 1e3, 1), "clone", round(ra["clone"]["events_ms_per_step"] * 1e3, 1))
  for k, v in (l.get("configs") or {}).items():
    print("  ", k, round(v["value"] / 1e6, 1), "M", round(v["avg_launch_ms"] * 1e3, 1), "us", round(v["frac"], 3), "of box fill", round(v["box_fill"]["frac_of_box_fill"], 3), p(v.get("plan")), v["placement"].get("dry_launch_us"), v["placement"].get("setup_s"))
  print("  cpu_baseline", l.get("cpu_baseline"))
PY
