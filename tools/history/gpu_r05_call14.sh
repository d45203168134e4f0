#!/bin/bash
# round 5, call 14: the round's final state — the whole GPU suite, smoke(), the profile set
# (tools/profile_round.sh r05: bench lines, kernel traces, HBM traffic passes), the bench line with
# the driver's flags, and bench lines of the two new levels
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call14; mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -14 $O/pytest.log
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 1500 bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; echo "profile rc=$?"; tail -5 $O/profile_round.log | cut -c1-400
( timeout 300 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "driver-flags rc=$?"
for s in coop_mining gift_refinements; do
  for obs in world agents; do
    timeout 300 python bench.py --substrate $s --obs $obs --no-cpu-baseline --no-substrate-api --no-rollout-api > $O/bench_${s}_$obs.json 2> $O/bench_${s}_$obs.err; echo "$s $obs rc=$?"
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_call14/bench_*.json")):
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], l["value"], l["ms_per_step"], l["roofline"]["frac"], l["config"]["workload"][:60])
    except Exception as e:
        print(f, "unreadable", e)
PY
