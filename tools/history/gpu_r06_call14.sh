#!/bin/bash
# round 6, call 14: the new resolve + mp_tune's pause search as the product: the bench line (driver's flags) on the
# product and on the old resolve (same box, one after the other, twice), then the whole GPU suite and smoke
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call14; mkdir -p $O
for rep in 1 2; do for lib in "" v1; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_$lib.so
  ( time MP_ENGINE_LIB=$L timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_${lib:-new}_$rep.json 2> $O/bench_${lib:-new}_$rep.err; echo "bench ${lib:-new} $rep rc=$?"
done; done
python - <<'PY'
import json
for f in ("bench_new_1", "bench_v1_1", "bench_new_2", "bench_v1_2"):
  l = json.loads(open(f"gpurun_out/r06_call14/{f}.json").read().strip().splitlines()[-1])
  print(f, "headline", round(l["value"] / 1e6, 1), "M", round(l["roofline"]["avg_launch_ms"] * 1e3, 1), "us", round(l["roofline"]["frac"], 3), "traffic", l["roofline"]["traffic"], "plan", l["plan"])
  print("  placement", l["placement"])
  print("  box_fill", json.dumps(l.get("box_fill")))
  sa = l.get("substrate_api") or {}
  print("  substrate_api", round(sa.get("avg_launch_ms", 0) * 1e3, 1), "us", round(sa.get("frac", 0), 3), sa.get("plan"))
  ra = (l.get("rollout_api") or {})
  print("  rollout_api", json.dumps({k: (round(v.get("avg_launch_ms", 0) * 1e3, 1) if isinstance(v, dict) else v) for k, v in ra.items()})[:300])
  for k, v in (l.get("configs") or {}).items():
    print("  ", k, round(v["value"] / 1e6, 1), "M", round(v["avg_launch_ms"] * 1e3, 1), "us", round(v["frac"], 3), "of box fill", round(v["box_fill"]["frac_of_box_fill"], 3), v.get("plan"), v["placement"].get("kind"))
PY
( time timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -12 $O/pytest_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-160
