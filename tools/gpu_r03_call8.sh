#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=300 2>&1 | tail -3
bash tools/gpu_ab.sh "- ntnone ntlate0" "" 2
echo "== draw-only (two launches): plain vs nt"
bash tools/gpu_ab.sh "- ntall" "--unfused" 1
