#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/pmc_sq_r02.sh commons_agents "--substrate commons_harvest__open --obs agents" 2>&1 | tail -3
head -30 gpurun_out/sq_commons_agents.md
