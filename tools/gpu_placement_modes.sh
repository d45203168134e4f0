#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$GRAFT_REPO_ROOT
(for m in many_buffers rebind new_engine_old_obs strides; do timeout 120 python tools/gpu_placement.py commons_harvest__open 4096 $m; done) 2>&1 | grep -v amdgpu.ids > $O/placement_modes.txt
cat $O/placement_modes.txt
