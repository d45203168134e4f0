#!/bin/bash
# round 4 (second session): the ring check under the product's head, then old / new library
# with the tuner on (what a caller gets), same buffers
set -u
out=gpurun_out/r04_head; mkdir -p $out
export PYTHONPATH=.
timeout 600 python tools/gpu_r04_ring_check.py > $out/ring_check_product.txt 2>&1; echo "ring check rc $?"; grep "FAIL\|failures" $out/ring_check_product.txt | head
export NBUF=5 MAPPED=3
for sub in "clean_up 4096 world" "clean_up 4096 both" "commons_harvest__open 4096 agents" "territory__rooms 8192 agents"; do
  n=$(echo $sub | tr ' ' '_')
  timeout 300 python tools/gpu_paired_ab.py $sub old - old:static_pct=100 -:static_pct=100 old > $out/tuned_$n.txt 2>&1; echo "rc $?"
  grep -v amdgpu.ids $out/tuned_$n.txt
done
