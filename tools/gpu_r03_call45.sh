#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
(timeout 120 python tools/gpu_placement.py commons_harvest__open 4096 probe_corr
VIEW=world timeout 120 python tools/gpu_placement.py clean_up 4096 probe_corr
timeout 120 python tools/gpu_placement.py territory__rooms 8192 probe_corr
VIEW=world timeout 120 python tools/gpu_placement.py clean_up 4096 probe_corr) 2>&1 | grep probe > $O/probe_corr.txt
cat $O/probe_corr.txt
