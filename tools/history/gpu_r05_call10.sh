#!/bin/bash
# round 5, call 10: does the record's STRIDE matter?  (MpDevOptions.record_pad: 64-byte blocks
# behind every record; same library, same buffers)
export TMPDIR=/tmp PYTHONPATH=.; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call10; mkdir -p $O
cd $R
for cfg in "commons_harvest__open 4096 agents" "clean_up 4096 world" "clean_up 4096 both" "territory__rooms 4096 agents"; do
  NBUF=2 MAPPED=4 timeout 300 python tools/gpu_paired_ab.py $cfg -:static_pct=100 -:static_pct=100,record_pad=1 -:static_pct=100,record_pad=2 -:static_pct=100,record_pad=3 -:static_pct=100,record_pad=5 -:static_pct=100 2>&1 | grep -v amdgpu.ids | tee -a $O/paired.txt
done
