#!/bin/bash
# round 5, call 35: per-agent views where batches of 3 leave workgroups idle (4096 / 3 -> 228, 8192 / 3 -> 249): territory, a matrix arena, clean_up's agents
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=.; O=gpurun_out/r05_both; mkdir -p $O
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py territory__rooms 8192 agents - -:batch_worlds=4,feeders=4 -:batch_worlds=4,feeders=2 -:batch_worlds=2,feeders=2 -:batch_worlds=2,feeders=4 > $O/territory_agents.txt 2>&1; tail -4 $O/territory_agents.txt
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py prisoners_dilemma_in_the_matrix__arena 4096 agents - -:batch_worlds=4,feeders=4 -:batch_worlds=4,feeders=6 -:batch_worlds=2,feeders=4 -:batch_worlds=2,feeders=6 > $O/pd_arena_agents.txt 2>&1; tail -4 $O/pd_arena_agents.txt
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - -:batch_worlds=2,feeders=4 -:batch_worlds=2,feeders=6 -:batch_worlds=4,feeders=4 > $O/commons_agents.txt 2>&1; tail -4 $O/commons_agents.txt
