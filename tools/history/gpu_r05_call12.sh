#!/bin/bash
# round 5, call 12: the same sources under other scheduling / optimisation flags, same buffers
# (libmp_engine_<tag>.so built with tools/ab_build.sh <tag> <flags>; "-" = the product build, -O3)
export TMPDIR=/tmp PYTHONPATH=.; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call12; mkdir -p $O
cd $R
V="-:static_pct=100 ilp:static_pct=100 memclause:static_pct=100 minreg:static_pct=100 prealloc:static_pct=100 o2:static_pct=100 relaxed:static_pct=100 -:static_pct=100"
for cfg in "clean_up 4096 world" "clean_up 4096 both" "commons_harvest__open 4096 agents" "territory__rooms 4096 agents"; do
  NBUF=1 MAPPED=5 timeout 400 python tools/gpu_paired_ab.py $cfg $V 2>&1 | grep -v amdgpu.ids | tee -a $O/paired.txt
done
