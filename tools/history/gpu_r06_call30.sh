#!/bin/bash
# round 6, call 30: the pause searched over all five values — does the tuner (dry probes here: 6 % margin) now reach what the
# forced plans reach on uneven buffers?  '-' = tuned, forced stock, forced 4 units, old resolve (tuned)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call30; mkdir -p $O
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents - -:static_pct=100 -:pace=5 -:batch_worlds=1,ring_batches=6,team=1,pace=3 v1 > $O/tuned_agents.txt 2>&1; grep -v amdgpu.ids $O/tuned_agents.txt | tail -10
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - -:static_pct=100 -:pace=5 -:batch_worlds=1,ring_batches=6,team=1,pace=3 v1 > $O/tuned_commons.txt 2>&1; grep -v amdgpu.ids $O/tuned_commons.txt | tail -10
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 both - -:static_pct=100 -:feeders=3,pace=3 -:batch_worlds=1,ring_batches=6,team=1,feeders=3 v1 > $O/tuned_both.txt 2>&1; grep -v amdgpu.ids $O/tuned_both.txt | tail -10
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 world - -:static_pct=100 -:batch_worlds=1,ring_batches=8,team=1,pace=3 v1 > $O/tuned_world.txt 2>&1; grep -v amdgpu.ids $O/tuned_world.txt | tail -10
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py territory__rooms 8192 agents - -:static_pct=100 -:pace=5 v1 > $O/tuned_territory.txt 2>&1; grep -v amdgpu.ids $O/tuned_territory.txt | tail -10
