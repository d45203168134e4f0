#!/bin/bash
# round 4: does mp_tune (dry launches) pick the plan that is fastest when really stepping?
# column 1 = the tuned engine (re-tuned on every buffer), then the three candidates forced
set -u
out=gpurun_out/r04_tune; mkdir -p $out
export NBUF=6 MAPPED=3 PYTHONPATH=.
timeout 300 python tools/gpu_paired_ab.py clean_up 4096 world - -:static_pct=100 \
  -:batch_worlds=1,ring_batches=8 -:batch_worlds=1,ring_batches=8,static_pct=50 > $out/clean_up_world.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - -:static_pct=100 \
  -:batch_worlds=1,ring_batches=6 -:batch_worlds=1,ring_batches=6,static_pct=50 > $out/commons_agents.txt 2>&1; echo "rc $?"
timeout 300 python tools/gpu_paired_ab.py territory__rooms 8192 agents - -:static_pct=100 \
  -:batch_worlds=1,ring_batches=6 -:batch_worlds=1,ring_batches=6,static_pct=50 > $out/territory_agents.txt 2>&1; echo "rc $?"
cat $out/*.txt
