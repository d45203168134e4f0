#!/bin/bash
# round 4: the tuner with its four candidates against the stock plan forced (commons_harvest, territory)
set -u
out=gpurun_out/r04_tune2; mkdir -p $out
timeout 600 python tools/gpu_r04_ring_check.py > $out/ring_check.log 2>&1; echo "ring_check rc $?"; tail -4 $out/ring_check.log
export NBUF=6 MAPPED=3 PYTHONPATH=.
timeout 300 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - -:static_pct=100 -:store_sc1=1 \
  -:batch_worlds=1,ring_batches=6 > $out/commons_agents.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py clean_up 4096 world - -:static_pct=100 > $out/clean_up_world.txt 2>&1; echo "rc $?"
cat $out/*.txt
