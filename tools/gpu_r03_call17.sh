#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
bash tools/gpu_ab1.sh "- waveprio copyprio" "" 3
