#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
(timeout 150 python tools/gpu_paired_ab.py clean_up 4096 world -
timeout 150 python tools/gpu_paired_ab.py clean_up 4096 world - -
timeout 150 python tools/gpu_paired_ab.py clean_up 4096 world - base
timeout 150 python tools/gpu_paired_ab.py clean_up 4096 world - - - -
timeout 150 python tools/gpu_paired_ab.py clean_up 4096 world - base t4 sc1) 2>&1 | grep -v amdgpu.ids | grep "mean\|builds" > $O/paired2.txt
cat $O/paired2.txt
