#!/bin/bash
# round 4 (second session): the start of a stepping launch (FramePlan::head) — parity under the
# ring check's plans, the variants on the same buffers (dev head = 1 + mask; the first column
# again at the end: order effects), the first 25 us of a frame
set -u
out=gpurun_out/r04_head; mkdir -p $out
export PYTHONPATH=.
if [ "${RING:-1}" = 1 ]; then
  for h in 1 10; do
    HEAD=$h timeout 600 python tools/gpu_r04_ring_check.py > $out/ring_check_head$h.txt 2>&1; echo "ring check (dev head $h) rc $?"
    grep "FAIL\|failures" $out/ring_check_head$h.txt | head -20
  done
fi
export NBUF=4 MAPPED=2
v=${1:-"-:head=1 -:head=2 -:head=6 -:head=10 -:head=12 -:head=16 -:head=1"}
timeout 300 python tools/gpu_paired_ab.py clean_up 4096 world $v > $out/clean_up_world.txt 2>&1; echo "rc $?"
grep -v amdgpu.ids $out/clean_up_world.txt
export MP_ENGINE_LIB=$PWD/meltingpot_amd/lib/libmp_engine_timeline.so UNTIL=25
for h in 1 10; do
  HEAD=$h timeout 120 python tools/gpu_timeline.py clean_up 4096 world > $out/timeline_head$h.txt 2>&1
  echo "=== head=$h"; grep -v amdgpu.ids $out/timeline_head$h.txt | head -14 | grep "span\|wave  0\|wave  8\|wave  9"
done
