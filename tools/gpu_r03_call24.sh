#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$GRAFT_REPO_ROOT
(timeout 120 python tools/gpu_episode_profile.py territory__rooms 8192 agents 1000
timeout 120 python tools/gpu_episode_profile.py commons_harvest__open 4096 agents 1000
timeout 120 python tools/gpu_episode_profile.py clean_up 4096 world 1000
timeout 120 python tools/gpu_episode_profile.py prisoners_dilemma_in_the_matrix__arena 8192 agents 1000
timeout 120 python tools/gpu_episode_profile.py clean_up 4096 agents 500) > $O/episode.txt 2>&1
cat $O/episode.txt
