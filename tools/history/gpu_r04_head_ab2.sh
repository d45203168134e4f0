#!/bin/bash
# round 4 (second session): a build against the previous one ("prev"), same buffers; ring check and step stamps first
set -u
out=gpurun_out/r04_head; mkdir -p $out
export PYTHONPATH=.
if [ "${RING:-1}" = 1 ]; then
  timeout 600 python tools/gpu_r04_ring_check.py > $out/ring_check_product.txt 2>&1; echo "ring check rc $?"; grep "FAIL\|failures" $out/ring_check_product.txt | head
  bash tools/gpu_step_timing.sh 2>&1 | grep "dispatch" | tail -2
fi
export NBUF=5 MAPPED=3
s=static_pct=100
for sub in "clean_up 4096 world" "clean_up 4096 both" "commons_harvest__open 4096 agents" "territory__rooms 8192 agents"; do
  n=$(echo $sub | tr ' ' '_')
  timeout 300 python tools/gpu_paired_ab.py $sub prev:$s -:$s prev:$s > $out/prev_$n.txt 2>&1; echo "rc $?"
  grep -v amdgpu.ids $out/prev_$n.txt | tail -12
done
