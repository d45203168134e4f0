#!/bin/bash
# round 6, call 12: per-agent views, new resolve: half the feeders (13 drawing waves, the tuner's fifth plan) x the pause
# between passes, against the old resolve tuned and with half the feeders.  Same buffers.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call12; mkdir -p $O
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents -:static_pct=100 -:pace=3 -:pace=5 -:feeders=3 -:feeders=3,pace=2 -:feeders=3,pace=3 -:feeders=3,pace=4 -:feeders=3,pace=5 -:feeders=3,pace=7 v1 v1:feeders=3 > $O/grid_agents.txt 2>&1; grep -v amdgpu.ids $O/grid_agents.txt | tail -11
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents -:static_pct=100 -:pace=3 -:pace=5 -:feeders=3 -:feeders=3,pace=2 -:feeders=3,pace=3 -:feeders=3,pace=4 -:feeders=3,pace=5 -:feeders=3,pace=7 v1 v1:feeders=3 > $O/grid_commons.txt 2>&1; grep -v amdgpu.ids $O/grid_commons.txt | tail -11
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 both -:static_pct=100 -:pace=2 -:feeders=3 -:feeders=3,pace=2 -:feeders=3,pace=3 -:feeders=3,pace=4 -:feeders=3,pace=5 v1 v1:feeders=3 > $O/grid_both.txt 2>&1; grep -v amdgpu.ids $O/grid_both.txt | tail -11
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py territory__rooms 8192 agents -:static_pct=100 -:pace=3 -:pace=5 -:pace=6 -:pace=7 v1 > $O/grid_territory.txt 2>&1; grep -v amdgpu.ids $O/grid_territory.txt | tail -11
