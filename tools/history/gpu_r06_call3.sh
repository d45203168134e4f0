#!/bin/bash
# round 6, call 3: the mushroom level with markings of their own position (the worlds of call 2's search replayed
# against the oracle), the tests the round touched, then the whole suite, smoke, and the bench line twice
# (driver's flags, defaults)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call3; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_mushroom.py -m gpu -x -q --durations=5 ) > $O/pytest_mushroom.log 2>&1; echo "mushroom tests rc=$?"; tail -8 $O/pytest_mushroom.log
( time timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -12 $O/pytest_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-160
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench (driver flags) rc=$?"; tail -3 $O/bench_driver_flags.err
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
for f in ("bench_driver_flags", "bench"):
  l = json.loads(open(f"gpurun_out/r06_call3/{f}.json").read().strip().splitlines()[-1])
  print(f, "headline", round(l["value"] / 1e6, 1), "M", round(l["roofline"]["avg_launch_ms"] * 1e3, 1), "us", round(l["roofline"]["frac"], 3), "traffic", l["roofline"]["traffic"], "plan", l["plan"])
  print("  placement", l["placement"])
  print("  box_fill", json.dumps(l.get("box_fill")))
  sa = l.get("substrate_api") or {}
  print("  substrate_api", round(sa.get("avg_launch_ms", 0) * 1e3, 1), "us", round(sa.get("frac", 0), 3), json.dumps((sa.get("box_fill") or {}).get("frac_of_box_fill")))
  for k, v in (l.get("configs") or {}).items():
    print("  ", k, round(v["value"] / 1e6, 1), "M", round(v["avg_launch_ms"] * 1e3, 1), "us", round(v["frac"], 3), "of box fill", round(v["box_fill"]["frac_of_box_fill"], 3), "bind_s", v["placement"].get("bind_s"), v["placement"].get("kind"))
PY
