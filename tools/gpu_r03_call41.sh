#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
(timeout 150 python tools/gpu_paired_ab.py clean_up 4096 world - base t3 t4 t5 t6 sc1 sc1t4
timeout 150 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - base t4 t6 sc1 sc1t4
WARM=300 timeout 150 python tools/gpu_paired_ab.py territory__rooms 8192 agents - t4 t6 sc1 sc1t4
timeout 150 python tools/gpu_paired_ab.py prisoners_dilemma_in_the_matrix__arena 8192 agents - t4 sc1t4) 2>&1 | grep -v amdgpu.ids > $O/paired.txt
cat $O/paired.txt
