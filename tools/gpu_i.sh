#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout -k 3 150 python -u tools/gpu_c.py 2>&1 | grep -v amdgpu.ids | tail -26
bash tools/gpu_f.sh 2>&1 | head -6
