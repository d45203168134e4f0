#!/bin/bash
# round 4: both views in one launch — how many of the renderer waves draw WORLD.RGB?
set -u
out=gpurun_out/r04_both; mkdir -p $out
export NBUF=4 MAPPED=2 PYTHONPATH=.
timeout 300 python tools/gpu_paired_ab.py clean_up 4096 both -:static_pct=100 -:world_waves=3 -:world_waves=4 \
  -:world_waves=5 -:world_waves=6 -:world_waves=4,feeders=4 -:world_waves=5,feeders=4 -:world_waves=4,feeders=3 \
  -:batch_worlds=1,ring_batches=6 -:batch_worlds=1,ring_batches=6,world_waves=4 > $out/clean_up_both.txt 2>&1; echo "rc $?"
cat $out/*.txt
