#!/bin/bash
# round 6, call 25: mp_tune with the feeders' late priority as a stage: the plan tests, then the bench line twice and the
# small levels' lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call25; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ring.py -m gpu -x -q -k "tuner or paced or teams or ring" --durations=5 ) > $O/pytest_plans.log 2>&1; echo "plan tests rc=$?"; tail -5 $O/pytest_plans.log
for rep in 1 2; do
  ( time timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "bench $rep rc=$?"; tail -3 $O/bench_$rep.err
done
python - <<'PY'
import json
for f in ("bench_1", "bench_2"):
  l = json.loads(open(f"gpurun_out/r06_call25/{f}.json").read().strip().splitlines()[-1])
  print(f, "headline", round(l["value"] / 1e6, 1), "M", round(l["roofline"]["avg_launch_ms"] * 1e3, 1), "us", round(l["roofline"]["frac"], 3), "plan", l["plan"])
  print("  placement", l["placement"]["dry_launch_us"], l["placement"]["setup_s"], "box_fill", l["box_fill"]["frac_of_box_fill"])
  sa = l.get("substrate_api") or {}
  print("  substrate_api", round(sa.get("avg_launch_ms", 0) * 1e3, 1), "us", round(sa.get("frac", 0), 3), sa.get("plan"))
  ra = l.get("rollout_api") or {}
  print("  rollout single", round(ra["single"]["events_ms_per_step"] * 1e3, 1), "us", ra["single"]["plan"], "ring", round(ra["ring"]["events_ms_per_step"] * 1e3, 1))
  for k, v in (l.get("configs") or {}).items():
    print("  ", k, round(v["value"] / 1e6, 1), "M", round(v["avg_launch_ms"] * 1e3, 1), "us", round(v["frac"], 3), "of box fill", round(v["box_fill"]["frac_of_box_fill"], 3), v.get("plan"), v["placement"].get("dry_launch_us"), v["placement"].get("setup_s"))
PY
for cfg in "--substrate externality_mushrooms__dense --obs agents" "--substrate coop_mining --obs agents" "--substrate collaborative_cooking__crowded --obs agents" "--substrate collaborative_cooking__cramped --obs agents"; do
  timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-traffic --no-substrate-api --no-rollout-api --no-steady-state --no-configs $cfg 2> /dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$cfg', round(l['value']/1e6,1), 'M', round(l['roofline']['avg_launch_ms']*1e3,1), 'us', round(l['roofline']['frac'],3), l['plan'])"
done
