#!/bin/bash
# round 6, call 23: the copy phase's bare road against the tested one (-DMP_NO_PLAIN_COPY), forced stock plans, same buffers;
# externality_mushrooms' per-agent view (call 22: 7 % behind the old resolve) under teams / pause / feeders
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call23; mkdir -p $O
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 world -:static_pct=100 tested:static_pct=100 -:batch_worlds=1,ring_batches=8,team=1 tested:batch_worlds=1,ring_batches=8,team=1 v1 > $O/copy_world.txt 2>&1; grep -v amdgpu.ids $O/copy_world.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents -:static_pct=100 tested:static_pct=100 -:feeders=3 tested:feeders=3 v1 > $O/copy_agents.txt 2>&1; grep -v amdgpu.ids $O/copy_agents.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents -:static_pct=100 tested:static_pct=100 -:feeders=3 tested:feeders=3 v1 > $O/copy_commons.txt 2>&1; grep -v amdgpu.ids $O/copy_commons.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py externality_mushrooms__dense 4096 agents - -:static_pct=100 -:pace=2 -:pace=3 -:pace=4 -:batch_worlds=1,ring_batches=8 -:batch_worlds=1,ring_batches=8,team=1 -:batch_worlds=1,ring_batches=8,team=1,pace=3 -:waves=14 -:feeders=6,waves=16 v1 > $O/mushrooms.txt 2>&1; grep -v amdgpu.ids $O/mushrooms.txt | tail -10
