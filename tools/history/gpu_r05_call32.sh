#!/bin/bash
# round 5, call 32: WORLD.RGB alone on the new levels: the stock plan (4 : 4, 12 waves) against others on the same buffers
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=.; O=gpurun_out/r05_both; mkdir -p $O
for s in externality_mushrooms__dense gift_refinements coop_mining clean_up; do
  NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py $s 4096 world - -:batch_worlds=4,feeders=4,waves=16 -:batch_worlds=6,feeders=6,waves=16 -:batch_worlds=8,feeders=4,waves=16 -:batch_worlds=8,feeders=4,waves=12 > $O/${s}_world.txt 2>&1
  tail -10 $O/${s}_world.txt
done
