#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
bash tools/gpu_ab.sh "- pf sc1 pfsc1 nt pfnt lateprio0 ablfeed" "" 2
