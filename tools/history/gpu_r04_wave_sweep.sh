#!/bin/bash
# round 4: waves x feeders on the single-world ring (clean_up WORLD.RGB), same buffers
set -u
out=gpurun_out/r04_waves; mkdir -p $out
export NBUF=5 MAPPED=3 PYTHONPATH=.
timeout 300 python tools/gpu_paired_ab.py clean_up 4096 world -:static_pct=100 \
  -:batch_worlds=1,ring_batches=8 -:batch_worlds=1,ring_batches=8,waves=10,feeders=2 \
  -:batch_worlds=1,ring_batches=8,waves=10,feeders=4 -:batch_worlds=1,ring_batches=8,waves=14,feeders=4 \
  -:batch_worlds=1,ring_batches=8,waves=16,feeders=4 -:batch_worlds=1,ring_batches=8,waves=16,feeders=8 \
  -:batch_worlds=1,ring_batches=8,waves=12,feeders=2 -:batch_worlds=1,ring_batches=4,waves=12,feeders=4 \
  -:batch_worlds=1,ring_batches=12,waves=12,feeders=4 -:batch_worlds=1,ring_batches=16,waves=12,feeders=4 > $out/clean_up_world.txt 2>&1; echo "rc $?"
cat $out/*.txt
