#!/bin/bash
# round 6, call 9: with the new resolve the renderers' count is the throttle (call 8: WORLD.RGB with 7 renderers instead of 8
# is flat at 98 - 103 us where 8 give 89 - 113).  The landscape over wave counts, same buffers, per config.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call9; mkdir -p $O
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 world - -:static_pct=100 -:waves=9 -:waves=10 -:waves=11 -:waves=13 -:waves=11,feeders=3 -:waves=10,feeders=3 -:waves=12,feeders=3 v1 > $O/sweep_world.txt 2>&1; grep -v amdgpu.ids $O/sweep_world.txt | tail -11
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents - -:static_pct=100 -:waves=13 -:waves=14 -:waves=15 -:waves=15,feeders=5 -:waves=14,feeders=4 -:waves=16,feeders=8 v1 > $O/sweep_agents.txt 2>&1; grep -v amdgpu.ids $O/sweep_agents.txt | tail -11
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - -:static_pct=100 -:waves=13 -:waves=14 -:waves=15 -:waves=15,feeders=5 -:waves=14,feeders=4 -:waves=16,feeders=8 v1 > $O/sweep_commons.txt 2>&1; grep -v amdgpu.ids $O/sweep_commons.txt | tail -11
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py territory__rooms 8192 agents - -:static_pct=100 -:waves=13 -:waves=14 -:waves=15 -:waves=16,feeders=4 -:waves=14,feeders=2 v1 > $O/sweep_territory.txt 2>&1; grep -v amdgpu.ids $O/sweep_territory.txt | tail -11
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 both - -:static_pct=100 -:waves=14 -:waves=15 -:feeders=3 -:waves=15,feeders=5 v1 > $O/sweep_both.txt 2>&1; grep -v amdgpu.ids $O/sweep_both.txt | tail -10
