#!/bin/bash
# One bench config: bench line + kernel trace + the two HBM-traffic PMC passes (as profile_round.sh).
# usage (on the GPU box): tools/profile_one.sh <tag> <name> "<bench args>"  -> gpurun_out/prof_<tag>/<name>.{md,bench.json}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1; mkdir -p $O; cd /tmp
name=$2; args=$3
timeout 200 python $R/bench.py $args > $O/$name.bench.json 2> $O/$name.bench.err
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $O/${name}_trace -o r -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-traffic $args > $O/${name}_trace.log 2>&1
echo "$name trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 150 rocprofv3 --pmc $c -d $O/${name}_$c -o r -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic $args > $O/${name}_$c.log 2>&1
  echo "$name $c rc=$?"
done
python3 $R/tools/rocprof_summary.py --trace $O/${name}_trace/r_results.db \
    --pmc $O/${name}_FETCH_SIZE/r_results.db $O/${name}_WRITE_SIZE/r_results.db \
    --out $O/$name.md --title "$1: $name (bench.py $args)"
rm -rf $O/${name}_trace $O/${name}_FETCH_SIZE $O/${name}_WRITE_SIZE
tail -c 600 $O/$name.bench.json
