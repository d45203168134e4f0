#!/bin/bash
# Round profile: for each bench config, a kernel trace and the two HBM-traffic PMC passes
# (one counter per pass, MI355X_MICROARCH.md), every rocprofv3 run under its own timeout;
# and the kernel trace of the `substrate_api` leg (both views bound: one dispatch a step).
# usage (on the GPU box): tools/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1; rm -rf $O; mkdir -p $O; cd /tmp
declare -A CFG
CFG[clean_up_world]=""
CFG[commons_agents]="--substrate commons_harvest__open --obs agents"
CFG[territory_agents]="--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5 --warmup 300"
for name in clean_up_world commons_agents territory_agents; do
  args=${CFG[$name]}
  timeout 300 python $R/bench.py $args > $O/$name.bench.json 2> $O/$name.bench.err
  timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $O/${name}_trace -o r -- python $R/bench.py --no-cpu-baseline --no-traffic --no-substrate-api --no-rollout-api --no-steady-state --no-configs --no-box-fill $args --steps 100 > $O/${name}_trace.log 2>&1
  echo "$name trace rc=$?"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 150 rocprofv3 --pmc $c -d $O/${name}_$c -o r -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-substrate-api --no-rollout-api --no-steady-state --no-configs --no-box-fill --place 1 $args > $O/${name}_$c.log 2>&1
    echo "$name $c rc=$?"
  done
  python3 $R/tools/rocprof_summary.py --trace $O/${name}_trace/r_results.db \
      --pmc $O/${name}_FETCH_SIZE/r_results.db $O/${name}_WRITE_SIZE/r_results.db \
      --last 100 --bench-log $O/${name}_trace.log --out $O/$name.md --title "$1: $name (bench.py $args)"
  rm -rf $O/${name}_trace $O/${name}_FETCH_SIZE $O/${name}_WRITE_SIZE
done
# the drop-in surface: the LAST 100 dispatches of this trace are substrate_api's steps
name=clean_up_substrate_api
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/${name}_trace -o r -- python $R/bench.py --no-cpu-baseline --no-traffic --no-rollout-api --no-steady-state --no-configs --no-box-fill --steps 100 > $O/${name}_trace.log 2>&1
echo "$name trace rc=$?"
python3 $R/tools/rocprof_summary.py --trace $O/${name}_trace/r_results.db --last 100 --bench-log $O/${name}_trace.log \
    --bench-key substrate_api --last-kernel ", 2>" --out $O/$name.md --title "$1: $name (substrate.build('clean_up', ..., num_worlds=4096): both views + six scalar kinds bound)"
rm -rf $O/${name}_trace
tail -c 2500 $O/clean_up_world.bench.json
