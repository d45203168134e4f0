#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
bash tools/gpu_ab.sh "- lateprio0 lateprio1 ablstep ablfeed" "" 2
