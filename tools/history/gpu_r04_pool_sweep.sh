#!/bin/bash
# round 4: ring depth x batch size x owned share, every plan timed on the same buffers
set -u
out=gpurun_out/r04_pool; mkdir -p $out
export NBUF=6 MAPPED=3 PYTHONPATH=.
timeout 300 python tools/gpu_paired_ab.py clean_up 4096 world - \
  -:batch_worlds=1,ring_batches=8 -:batch_worlds=1,ring_batches=8,static_pct=75 \
  -:batch_worlds=1,ring_batches=8,static_pct=50 -:batch_worlds=1,ring_batches=8,static_pct=25 \
  -:batch_worlds=2,ring_batches=4 -:batch_worlds=2,ring_batches=4,static_pct=75 -:batch_worlds=2,ring_batches=4,static_pct=50 \
  -:static_pct=75 -:static_pct=50 > $out/clean_up_world.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - \
  -:batch_worlds=1,ring_batches=6 -:batch_worlds=1,ring_batches=6,static_pct=75 -:batch_worlds=1,ring_batches=6,static_pct=50 \
  -:static_pct=67 -:static_pct=50 -:batch_worlds=2,ring_batches=3,static_pct=50 > $out/commons_agents.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py territory__rooms 8192 agents - \
  -:batch_worlds=1,ring_batches=6 -:batch_worlds=1,ring_batches=6,static_pct=75 -:batch_worlds=1,ring_batches=6,static_pct=50 \
  -:static_pct=75 -:static_pct=50 > $out/territory_agents.txt 2>&1; echo "rc $?"
cat $out/*.txt
