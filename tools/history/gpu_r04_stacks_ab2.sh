#!/bin/bash
set -u
out=gpurun_out/r04_stacks2; mkdir -p $out
export NBUF=6 MAPPED=6 PYTHONPATH=.
timeout 300 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents -:static_pct=100 -:no_stacks=1 \
  -:batch_worlds=1,ring_batches=6 -:batch_worlds=1,ring_batches=6,no_stacks=1 > $out/commons_agents.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py territory__rooms 8192 agents -:static_pct=100 -:no_stacks=1 \
  -:batch_worlds=1,ring_batches=6 -:batch_worlds=1,ring_batches=6,no_stacks=1 > $out/territory_agents.txt 2>&1; echo "rc $?"
cat $out/*.txt
