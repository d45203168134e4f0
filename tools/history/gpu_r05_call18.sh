#!/bin/bash
# round 5, call 18: the beams' lane / n and 64 / n from the host (BeamShape.per / .magic) against
# the previous library (libmp_engine_prev.so), same buffers
export TMPDIR=/tmp PYTHONPATH=.; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call18; mkdir -p $O
cd $R
for cfg in "clean_up 4096 world" "clean_up 4096 both" "commons_harvest__open 4096 agents" "territory__rooms 4096 agents"; do
  NBUF=1 MAPPED=7 timeout 300 python tools/gpu_paired_ab.py $cfg prev:static_pct=100 -:static_pct=100 prev:static_pct=100 -:static_pct=100 2>&1 | grep -v amdgpu.ids | tee -a $O/paired.txt
done
