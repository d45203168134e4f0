#!/bin/bash
# round 6, call 38: per-phase cycle counts of a clean_up step (-DMP_STEP_TIMING, the stand-alone step kernel)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call38; mkdir -p $O
bash tools/gpu_step_timing.sh 2>&1 | tee $O/step_timing.txt | tail -24
