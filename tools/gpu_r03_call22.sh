#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1
bash tools/gpu_ab.sh "base -" "" 5 > $O/ab2.txt 2>&1; cat $O/ab2.txt
