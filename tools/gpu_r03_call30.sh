#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$GRAFT_REPO_ROOT
(for m in strides strides; do timeout 120 python tools/gpu_bimodal3.py commons_harvest__open 4096 $m; done) 2>&1 | grep -v amdgpu.ids > $O/bimodal6.txt
cat $O/bimodal6.txt
