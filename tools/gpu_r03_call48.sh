#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
(timeout 150 python tools/gpu_paired_ab.py clean_up 4096 world - -:waves=14 -:waves=16 -:waves=16,feeders=8 -:waves=16,batch_worlds=3,feeders=6 -:waves=13,batch_worlds=3,feeders=3 -:waves=10
timeout 150 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - -:feeders=3 -:batch_worlds=4,feeders=4 -:batch_worlds=2,feeders=4
WARM=300 timeout 150 python tools/gpu_paired_ab.py territory__rooms 8192 agents - -:feeders=2 -:feeders=6 -:batch_worlds=2,feeders=4) 2>&1 | grep -v amdgpu.ids > $O/paired_plans.txt
cat $O/paired_plans.txt
