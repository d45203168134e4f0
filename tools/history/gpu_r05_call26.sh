#!/bin/bash
# round 5, call 26: kernel trace + HBM traffic of the ninth level's fused launch (profile_round.sh's recipe)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r05_mush; rm -rf $O; mkdir -p $O; cd /tmp
name=externality_mushrooms_agents
args="--substrate externality_mushrooms__dense --obs agents"
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $O/${name}_trace -o r -- python $R/bench.py --no-cpu-baseline --no-traffic --no-substrate-api --no-rollout-api --no-steady-state $args --steps 100 > $O/${name}_trace.log 2>&1
echo "$name trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 150 rocprofv3 --pmc $c -d $O/${name}_$c -o r -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --no-substrate-api --no-rollout-api --no-steady-state --place 1 $args > $O/${name}_$c.log 2>&1
  echo "$name $c rc=$?"
done
python3 $R/tools/rocprof_summary.py --trace $O/${name}_trace/r_results.db \
    --pmc $O/${name}_FETCH_SIZE/r_results.db $O/${name}_WRITE_SIZE/r_results.db \
    --last 100 --bench-log $O/${name}_trace.log --out $O/$name.md --title "r05: $name (bench.py $args)"
rm -rf $O/${name}_trace $O/${name}_FETCH_SIZE $O/${name}_WRITE_SIZE
head -30 $O/$name.md
