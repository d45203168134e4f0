#!/bin/bash
# round 5, call 9: the build that leaves the next step's orders in the record against the
# previous one on the same buffers (tools/gpu_paired_ab.py; libmp_engine_old.so = the library of
# commit 7e7e50f), and the head of a frame in both (timeline builds, tools/gpu_timeline.py)
export TMPDIR=/tmp PYTHONPATH=.; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call9; mkdir -p $O
cd $R
for cfg in "clean_up 4096 world" "clean_up 4096 both" "commons_harvest__open 4096 agents" "territory__rooms 4096 agents"; do
  NBUF=2 MAPPED=6 timeout 300 python tools/gpu_paired_ab.py $cfg old:static_pct=100 -:static_pct=100 -:static_pct=100,no_next_orders=1 old:static_pct=100 2>&1 | grep -v amdgpu.ids | tee -a $O/paired.txt
done
for lib in oldtl timeline; do
  HEAD=2 UNTIL=30 MP_ENGINE_LIB=$PWD/meltingpot_amd/lib/libmp_engine_$lib.so timeout 120 python tools/gpu_timeline.py clean_up 4096 world > $O/tl_world_$lib.txt 2>&1; echo "$lib rc $?"
done
grep -c "" $O/tl_world_*.txt
