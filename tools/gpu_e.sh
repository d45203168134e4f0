#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
for mode in render fused; do timeout -k 3 25 python -u tools/gpu_d.py $mode 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-80; done
timeout -k 3 120 python -u tools/gpu_c.py 2>&1 | grep -v amdgpu.ids | tail -30
