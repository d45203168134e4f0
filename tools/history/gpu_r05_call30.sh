#!/bin/bash
# round 5, call 30: what substrate.build binds on clean_up (both views, 4096 worlds): the product's tuned plan against forced ones on the same buffers
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=.; O=gpurun_out/r05_both; mkdir -p $O
NBUF=4 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 both - -:batch_worlds=3,feeders=6 -:batch_worlds=3,feeders=3 -:batch_worlds=4,feeders=4 -:batch_worlds=4,feeders=3 -:batch_worlds=2,feeders=4 > $O/clean_up_both.txt 2>&1
cat $O/clean_up_both.txt | tail -12
NBUF=4 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents - -:batch_worlds=3,feeders=3 -:batch_worlds=4,feeders=4 -:batch_worlds=4,feeders=2 > $O/clean_up_agents.txt 2>&1
cat $O/clean_up_agents.txt | tail -10
