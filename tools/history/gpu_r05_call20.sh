#!/bin/bash
# round 5, call 20: small substrates with WORLD.RGB or both views bound (what substrate.build binds):
# the stock plan (batches of 4, 4 feeders) against batches of 8 with 8 feeders, same buffers
export TMPDIR=/tmp PYTHONPATH=.; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call20; mkdir -p $O
cd $R
for s in collaborative_cooking__cramped prisoners_dilemma_in_the_matrix__repeated coins collaborative_cooking__crowded; do
  for v in world both; do
    NBUF=1 MAPPED=3 timeout 300 python tools/gpu_paired_ab.py $s 4096 $v -:static_pct=100 -:static_pct=100,feeders=8,batch_worlds=8 -:static_pct=100,feeders=6,batch_worlds=6 -:static_pct=100,feeders=8,batch_worlds=4 2>&1 | grep -E "x4096|mean" | tee -a $O/paired.txt
  done
done
