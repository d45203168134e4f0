#!/bin/bash
# round 5, call 33: WORLD.RGB alone, 16 - 64 KB a world (the larger kitchens, externality_mushrooms): stock against batches of 6 / 8 with 16 waves
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=.; O=gpurun_out/r05_both; mkdir -p $O
for s in collaborative_cooking__crowded collaborative_cooking__figure_eight externality_mushrooms__dense; do
  NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py $s 4096 world - -:batch_worlds=6,feeders=6,waves=16 -:batch_worlds=8,feeders=8,waves=16 -:batch_worlds=8,feeders=4,waves=16 -:batch_worlds=6,feeders=3,waves=16 > $O/${s}_world2.txt 2>&1
  tail -4 $O/${s}_world2.txt
done
