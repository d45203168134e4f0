#!/bin/bash
# round 5, call 19: the profile set and the bench lines of the round's final library
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call19; mkdir -p $O
cd $R
timeout 1500 bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1; echo "profile rc=$?"
( timeout 300 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "driver-flags rc=$?"
for s in coop_mining gift_refinements collaborative_cooking__crowded collaborative_cooking__cramped; do
  for obs in world agents; do
    timeout 300 python bench.py --substrate $s --obs $obs --no-cpu-baseline --no-substrate-api --no-rollout-api > $O/bench_${s}_$obs.json 2> $O/bench_${s}_$obs.err; echo "$s $obs rc=$?"
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_call19/bench_*.json")) + sorted(glob.glob("gpurun_out/prof_r05/*.bench.json")):
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(l["value"]/1e6,1), round(l["ms_per_step"]*1e3,2), round(l["roofline"]["avg_launch_ms"]*1e3,2), round(l["roofline"]["frac"],3))
    except Exception as e:
        print(f, "unreadable", e)
PY
