#!/bin/bash
# round 5, call 11: the kernel arguments asked for by four renderer waves only, in the order
# the feeders need them (libmp_engine_warm4.so) against every wave asking for all of them at
# its first instruction ("-"), same buffers; the head of a frame in the new form
export TMPDIR=/tmp PYTHONPATH=.; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call11; mkdir -p $O
cd $R
for cfg in "clean_up 4096 world" "clean_up 4096 both" "commons_harvest__open 4096 agents"; do
  NBUF=2 MAPPED=6 timeout 300 python tools/gpu_paired_ab.py $cfg -:static_pct=100 warm4:static_pct=100 -:static_pct=100 warm4:static_pct=100 2>&1 | grep -v amdgpu.ids | tee -a $O/paired.txt
done
HEAD=2 UNTIL=30 MP_ENGINE_LIB=$PWD/meltingpot_amd/lib/libmp_engine_warm4tl.so timeout 120 python tools/gpu_timeline.py clean_up 4096 world > $O/tl_world_warm4.txt 2>&1; echo "tl rc $?"
head -16 $O/tl_world_warm4.txt | cut -c1-200
