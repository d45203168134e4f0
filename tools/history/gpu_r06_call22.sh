#!/bin/bash
# round 6, call 22: the copy road / reciprocals (library "pre" = before them) on whatever box this is — call 19's had no even
# buffer — forced stock plans; and the small-view levels, old resolve against new
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call22; mkdir -p $O
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 world -:static_pct=100 pre:static_pct=100 -:batch_worlds=1,ring_batches=8,team=1 v1 > $O/copy_world.txt 2>&1; grep -v amdgpu.ids $O/copy_world.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents -:static_pct=100 pre:static_pct=100 -:feeders=3 pre:feeders=3 v1 > $O/copy_agents.txt 2>&1; grep -v amdgpu.ids $O/copy_agents.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents -:static_pct=100 pre:static_pct=100 -:feeders=3 pre:feeders=3 v1 > $O/copy_commons.txt 2>&1; grep -v amdgpu.ids $O/copy_commons.txt | tail -10
for sub in collaborative_cooking__cramped collaborative_cooking__crowded coins prisoners_dilemma_in_the_matrix__repeated externality_mushrooms__dense; do
  NBUF=2 MAPPED=2 timeout 300 python tools/gpu_paired_ab.py $sub 4096 agents - v1 > $O/small_$sub.txt 2>&1; grep -v amdgpu.ids $O/small_$sub.txt | tail -7
done
