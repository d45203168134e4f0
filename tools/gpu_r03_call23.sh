#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 120 python tools/gpu_bimodal.py territory__rooms 8192 4; done > $O/bimodal.txt 2>&1
for i in 1 2; do timeout 120 python tools/gpu_bimodal.py commons_harvest__open 4096 4; done >> $O/bimodal.txt 2>&1
cat $O/bimodal.txt
