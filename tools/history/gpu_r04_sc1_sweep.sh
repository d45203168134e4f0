#!/bin/bash
# round 4: sc1 pixel stores as a dimension of the plan, same buffers
set -u
out=gpurun_out/r04_sc1; mkdir -p $out
export NBUF=6 MAPPED=3 PYTHONPATH=.
timeout 300 python tools/gpu_paired_ab.py clean_up 4096 world -:static_pct=100 -:store_sc1=1 \
  -:batch_worlds=1,ring_batches=8 -:batch_worlds=1,ring_batches=8,store_sc1=1 > $out/clean_up_world.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents -:static_pct=100 -:store_sc1=1 \
  -:batch_worlds=1,ring_batches=6 -:batch_worlds=1,ring_batches=6,store_sc1=1 -:batch_worlds=1,ring_batches=6,static_pct=50,store_sc1=1 > $out/commons_agents.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py territory__rooms 8192 agents -:static_pct=100 -:store_sc1=1 \
  -:batch_worlds=1,ring_batches=6,static_pct=50 -:batch_worlds=1,ring_batches=6,static_pct=50,store_sc1=1 > $out/territory_agents.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py clean_up 4096 both -:static_pct=100 -:store_sc1=1 > $out/clean_up_both.txt 2>&1; echo "rc $?"
cat $out/*.txt
