#!/bin/bash
# round 6, call 10: a renderer wave that SLEEPS between two passes (FramePlan::pace, units of 512 cycles) — does the new
# resolve + a pause serve the bad buffers as the old, slower resolve did (105 us flat), and keep the good ones at 90 - 95?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call10; mkdir -p $O
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 world -:static_pct=100 -:pace=2 -:pace=3 -:pace=4 -:pace=5 -:pace=7 -:pace=9 -:pace=13 v1 > $O/pace_world.txt 2>&1; grep -v amdgpu.ids $O/pace_world.txt | tail -11
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents -:static_pct=100 -:pace=2 -:pace=3 -:pace=5 -:pace=7 -:pace=9 -:pace=13 v1 > $O/pace_agents.txt 2>&1; grep -v amdgpu.ids $O/pace_agents.txt | tail -11
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents -:static_pct=100 -:pace=2 -:pace=3 -:pace=5 -:pace=7 -:pace=9 -:pace=13 v1 > $O/pace_commons.txt 2>&1; grep -v amdgpu.ids $O/pace_commons.txt | tail -11
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py territory__rooms 8192 agents -:static_pct=100 -:pace=2 -:pace=3 -:pace=5 -:pace=7 -:pace=9 -:pace=13 v1 > $O/pace_territory.txt 2>&1; grep -v amdgpu.ids $O/pace_territory.txt | tail -11
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 both -:static_pct=100 -:pace=2 -:pace=3 -:pace=5 -:pace=7 -:pace=9 v1 > $O/pace_both.txt 2>&1; grep -v amdgpu.ids $O/pace_both.txt | tail -10
