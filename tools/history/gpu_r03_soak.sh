#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout 800 python tests/tools/soak.py 2500 24 2>&1 | grep -v amdgpu.ids | tail -30
timeout 300 python tests/tools/big_n.py clean_up 65536 2>&1 | grep -v amdgpu.ids | tail -6
