#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$GRAFT_REPO_ROOT
(for i in 1 2 3; do timeout 120 python tools/gpu_bimodal2.py commons_harvest__open 4096; done) 2>&1 | grep -v amdgpu.ids > $O/bimodal2.txt
cat $O/bimodal2.txt
