#!/bin/bash
# round 4 (second session): tickets taken a pass ahead (head bit 2) or not, remembered slot flags in both
set -u
out=gpurun_out/r04_head; mkdir -p $out
export PYTHONPATH=. NBUF=5 MAPPED=3
s=static_pct=100
for sub in "clean_up 4096 world" "clean_up 4096 both" "commons_harvest__open 4096 agents"; do
  n=$(echo $sub | tr ' ' '_')
  timeout 300 python tools/gpu_paired_ab.py $sub prev:$s -:$s -:$s,head=4 -:$s,head=4,batch_worlds=1,ring_batches=8 prev:$s,batch_worlds=1,ring_batches=8 prev:$s > $out/ahead_$n.txt 2>&1; echo "rc $?"
  grep -v amdgpu.ids $out/ahead_$n.txt | tail -12
done
