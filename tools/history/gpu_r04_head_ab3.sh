#!/bin/bash
# round 4 (second session): a build against the previous one on MANY buffers of the headline view
set -u
out=gpurun_out/r04_head; mkdir -p $out
export PYTHONPATH=. NBUF=8 MAPPED=10
s=static_pct=100
timeout 300 python tools/gpu_paired_ab.py clean_up 4096 world prev:$s -:$s prev:$s > $out/prev_world_many.txt 2>&1; echo "rc $?"
grep -v amdgpu.ids $out/prev_world_many.txt
