#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
(timeout 150 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - ragged
timeout 150 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents ragged -
timeout 150 python tools/gpu_paired_ab.py commons_harvest__closed 4096 agents - ragged
timeout 150 python tools/gpu_paired_ab.py clean_up 4096 agents - ragged
timeout 150 python tools/gpu_paired_ab.py prisoners_dilemma_in_the_matrix__arena 4096 agents - ragged) 2>&1 | grep -v amdgpu.ids > $O/ragged.txt
cat $O/ragged.txt
