#!/bin/bash
# round 4: cell stacks (a world's cells resolved once by its feeder) against every viewer
# resolving its window from the planes, same buffers; "-" = tuned product
set -u
out=gpurun_out/r04_stacks; mkdir -p $out
export NBUF=5 MAPPED=3 PYTHONPATH=.
timeout 300 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - -:static_pct=100 -:no_stacks=1 \
  -:batch_worlds=1,ring_batches=6 -:batch_worlds=1,ring_batches=6,no_stacks=1 > $out/commons_agents.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py territory__rooms 8192 agents - -:static_pct=100 -:no_stacks=1 \
  -:batch_worlds=1,ring_batches=6 -:batch_worlds=1,ring_batches=6,no_stacks=1 \
  -:batch_worlds=1,ring_batches=6,static_pct=50 -:batch_worlds=1,ring_batches=6,static_pct=50,no_stacks=1 > $out/territory_agents.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py clean_up 4096 agents - -:static_pct=100 -:no_stacks=1 > $out/clean_up_agents.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py prisoners_dilemma_in_the_matrix__arena 8192 agents - -:static_pct=100 -:no_stacks=1 > $out/pd_arena_agents.txt 2>&1; echo "rc $?"
cat $out/*.txt
