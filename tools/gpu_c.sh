#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout -k 5 280 python -u tools/gpu_c.py 2>&1 | grep -v amdgpu.ids | tail -60
