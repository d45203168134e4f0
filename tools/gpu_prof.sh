#!/bin/bash
# round profile on the GPU box: bench lines + kernel traces + HBM traffic (profile_round.sh), then SQ counters
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
bash tools/profile_round.sh r02
bash tools/pmc_sq_r02.sh clean_up_world ""
bash tools/pmc_sq_r02.sh territory_agents "--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5"
