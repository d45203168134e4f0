#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
(timeout 150 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - -:max_groups=205 -:max_groups=192 -:max_groups=171 -:max_groups=152 -:max_groups=128
timeout 150 python tools/gpu_paired_ab.py clean_up 4096 world - -:max_groups=228 -:max_groups=205 -:max_groups=171 -:max_groups=128
WARM=300 timeout 150 python tools/gpu_paired_ab.py territory__rooms 8192 agents - -:max_groups=228 -:max_groups=205 -:max_groups=171) 2>&1 | grep -v amdgpu.ids > $O/groups.txt
cat $O/groups.txt
