#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03c; mkdir -p $O
bash tools/gpu_tests.sh 300 900 2>&1 | tee $O/tests.txt
QUICK=1 timeout 200 tools/ubench/store_ceiling > $O/store_prio.md 2>&1; cat $O/store_prio.md
bash tools/gpu_ab.sh "- copyprio waveprio" "" 2 2>&1 | tee $O/ab_prio.txt
