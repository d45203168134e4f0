#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$GRAFT_REPO_ROOT
(timeout 120 python tools/gpu_episode_profile.py territory__rooms 8192 none 1000
timeout 120 python tools/gpu_episode_profile.py territory__rooms 8192 world 1000
timeout 120 python tools/gpu_episode_profile.py prisoners_dilemma_in_the_matrix__arena 8192 none 500
timeout 120 python tools/gpu_episode_profile.py territory__open 8192 agents 600
timeout 120 python tools/gpu_episode_profile.py territory__inside_out 8192 agents 600 ) 2>&1 | grep -v "mp_engine\|amdgpu.ids" > $O/episode2.txt
cat $O/episode2.txt
