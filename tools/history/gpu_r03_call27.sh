#!/bin/bash
# dev helper: every substrate, per-agent view, 4096 worlds, us per step by block of 50 steps over 500 steps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$GRAFT_REPO_ROOT
for s in $(ls meltingpot_amd/assets/*.mpk | xargs -n1 basename | sed 's/.mpk//'); do
  timeout 120 python tools/gpu_episode_profile.py $s 4096 agents 500 2>&1 | grep -v "amdgpu.ids\|frame plan"
done > $O/episode_all.txt
cat $O/episode_all.txt
