#!/bin/bash
# round 6, call 40: mp_place_output ranks its candidates by a quick look (stock plan + team order) and searches plans only
# on the one it keeps: the place / tune / ring tests, then the bench line twice (setup_s per leg)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call40; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ring.py tests/test_gpu_surface.py tests/test_substrate_api.py -m gpu -x -q -k "tune or place or ring or placement or probe or bind or rollout" --durations=5 ) > $O/pytest_place.log 2>&1; echo "place tests rc=$?"; tail -4 $O/pytest_place.log
for rep in 1 2; do
  ( time timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_$rep.json 2> $O/bench_$rep.err; echo "bench $rep rc=$?"; grep real $O/bench_$rep.err
done
python - <<'PY'
import json
for f in ("bench_1", "bench_2"):
  l = json.loads(open(f"gpurun_out/r06_call40/{f}.json").read().strip().splitlines()[-1])
  p = lambda d: {k: v for k, v in d.items() if k in ("batch_worlds", "feeders", "pace", "xcd_teams", "late_feeder_priority", "sc1_stores", "pooled_batches")}
  print(f, "headline", round(l["value"] / 1e6, 1), "M", round(l["roofline"]["avg_launch_ms"] * 1e3, 1), "us", round(l["roofline"]["frac"], 3), p(l["plan"]), l["placement"]["dry_launch_us"], "setup", l["placement"]["setup_s"])
  sa = l["substrate_api"]
  print("  substrate_api", round(sa["avg_launch_ms"] * 1e3, 1), round(sa["frac"], 3), p(sa["plan"]), {k: v["setup_s"] for k, v in sa["placement"].items()})
  ra = l["rollout_api"]
  print("  rollout single", round(ra["single"]["events_ms_per_step"] * 1e3, 1), p(ra["single"]["plan"]), ra["single"]["setup_s"], "ring", round(ra["ring"]["events_ms_per_step"] * 1e3, 1), ra["ring"]["setup_s"])
  for k, v in l["configs"].items():
    print("  ", k, round(v["avg_launch_ms"] * 1e3, 1), "us", round(v["frac"], 3), p(v["plan"]), v["placement"]["dry_launch_us"], "setup", v["placement"]["setup_s"])
PY
