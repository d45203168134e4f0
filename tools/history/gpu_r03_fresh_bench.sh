#!/bin/bash
# the bench lines on a box no earlier process has touched (profiles/r03_buffer_placement.md:
# the box's memory state decides how many fast placements Engine.place() finds), then the
# headline's kernel trace + traffic passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd /tmp
timeout 300 python $R/bench.py > $O/clean_up_world.json 2> $O/clean_up_world.err
timeout 200 python $R/bench.py --substrate commons_harvest__open --obs agents > $O/commons_agents.json 2>/dev/null
timeout 200 python $R/bench.py --substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5 --warmup 300 > $O/territory_agents.json 2>/dev/null
timeout 200 python $R/bench.py --no-cpu-baseline --no-traffic --substrate prisoners_dilemma_in_the_matrix__arena --obs agents --worlds 8192 > $O/pd_arena.json 2>/dev/null
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python $R/bench.py --no-cpu-baseline --no-traffic --steps 100 > $O/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 5 150 rocprofv3 --pmc $c -d $O/$c -o r -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-traffic --place 1 > $O/$c.log 2>&1; done
python3 $R/tools/rocprof_summary.py --trace $O/trace/r_results.db --pmc $O/FETCH_SIZE/r_results.db $O/WRITE_SIZE/r_results.db --last 100 --bench-log $O/trace.log --out $O/clean_up_world.md --title "r03, fresh box: clean_up_world (bench.py)" > /dev/null
rm -rf $O/trace $O/FETCH_SIZE $O/WRITE_SIZE
for f in clean_up_world commons_agents territory_agents pd_arena; do tail -1 $O/$f.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$f', round(d['value']/1e6,1), 'M', round(d['kernels_ms']['frame']*1e3,1), 'us frac', round(d['roofline']['frac'],3), d['placement'])"; done
grep -A2 "last 100\|inside this traced" $O/clean_up_world.md | head -8
