#!/bin/bash
# round 5, call 31: both views bound (what substrate.build binds) on the five- and six-viewer levels: the tuned plan against forced ones, same buffers
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=.; O=gpurun_out/r05_both; mkdir -p $O
for s in externality_mushrooms__dense gift_refinements coop_mining; do
  NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py $s 4096 both - -:batch_worlds=3,feeders=3 -:batch_worlds=4,feeders=4 -:batch_worlds=4,feeders=6 -:batch_worlds=2,feeders=4 > $O/${s}_both.txt 2>&1
  tail -10 $O/${s}_both.txt
done
