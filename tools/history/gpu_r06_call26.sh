#!/bin/bash
# round 6, call 26: mp_tune's second look (the close candidates timed three times as long): tune / place / ring tests,
# then the bench line three times — does the headline's plan still change from run to run?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call26; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ring.py tests/test_gpu_surface.py -m gpu -x -q -k "tune or place or ring or paced or teams or placement or probe" --durations=5 ) > $O/pytest_plans.log 2>&1; echo "plan tests rc=$?"; tail -5 $O/pytest_plans.log
for rep in 1 2 3; do
  ( time timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "bench $rep rc=$?"; tail -3 $O/bench_$rep.err | grep real
done
python - <<'PY'
import json
for f in ("bench_1", "bench_2", "bench_3"):
  l = json.loads(open(f"gpurun_out/r06_call26/{f}.json").read().strip().splitlines()[-1])
  p = lambda d: {k: v for k, v in d.items() if k in ("batch_worlds", "feeders", "pace", "xcd_teams", "late_feeder_priority", "sc1_stores", "pooled_batches")}
  print(f, "headline", round(l["value"] / 1e6, 1), "M", round(l["roofline"]["avg_launch_ms"] * 1e3, 1), "us", round(l["roofline"]["frac"], 3), "plan", p(l["plan"]))
  print("  placement", l["placement"]["dry_launch_us"], l["placement"]["setup_s"], "box_fill", round(l["box_fill"]["frac_of_box_fill"], 3))
  sa = l.get("substrate_api") or {}
  print("  substrate_api", round(sa.get("avg_launch_ms", 0) * 1e3, 1), "us", round(sa.get("frac", 0), 3), p(sa.get("plan")))
  ra = l.get("rollout_api") or {}
  print("  rollout single", round(ra["single"]["events_ms_per_step"] * 1e3, 1), "us", p(ra["single"]["plan"]), "ring", round(ra["ring"]["events_ms_per_step"] * 1e3, 1), "setup", ra["ring"].get("setup_s"))
  for k, v in (l.get("configs") or {}).items():
    print("  ", k, round(v["value"] / 1e6, 1), "M", round(v["avg_launch_ms"] * 1e3, 1), "us", round(v["frac"], 3), "of box fill", round(v["box_fill"]["frac_of_box_fill"], 3), p(v.get("plan")), v["placement"].get("dry_launch_us"), v["placement"].get("setup_s"))
PY
