#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$GRAFT_REPO_ROOT
(timeout 120 python tools/gpu_episode_profile.py territory__rooms 8192 agents 1000
timeout 120 python tools/gpu_episode_profile.py territory__rooms 8192 world 600
timeout 120 python tools/gpu_episode_profile.py territory__open 8192 agents 600
timeout 120 python tools/gpu_episode_profile.py prisoners_dilemma_in_the_matrix__arena 8192 agents 500
timeout 120 python tools/gpu_episode_profile.py commons_harvest__open 4096 agents 300
timeout 120 python tools/gpu_episode_profile.py clean_up 4096 world 300) 2>&1 | grep -v "amdgpu.ids\|frame plan" > $O/episode3.txt
cat $O/episode3.txt
for f in tests/test_gpu_parity.py tests/test_gpu_matrix.py tests/test_gpu_surface.py; do
timeout -k 10 600 python -u -m pytest $f -m gpu -q -x --timeout=240 --timeout-method=thread > $O/$(basename $f .py).log 2>&1
echo "$f rc=$? : $(tail -1 $O/$(basename $f .py).log)"; done
