#!/bin/bash
# round 6, call 11: the renderers alone (draw-only launch) against the fused step, old and new resolve, same buffers:
# where does the new resolve lose on the per-agent views?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call11; mkdir -p $O
for cfg in "clean_up 4096 world" "clean_up 4096 agents" "commons_harvest__open 4096 agents" "territory__rooms 8192 agents"; do
  timeout 600 python tools/gpu_draw_ab.py $cfg - v1 > $O/draw_$(echo $cfg | tr ' ' '_').txt 2>&1; grep -v amdgpu.ids $O/draw_$(echo $cfg | tr ' ' '_').txt | tail -8
done
NBUF=2 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents -:static_pct=100 -:late_feeder_prio=2 -:late_feeder_prio=4 -:feeders=3 -:feeders=4 v1 v1:feeders=3 > $O/prio_agents.txt 2>&1; grep -v amdgpu.ids $O/prio_agents.txt | tail -9
