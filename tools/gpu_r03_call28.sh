#!/bin/bash
# dev helper: plan sweeps after the renderer's instruction diet (late-episode state: --warmup 300 for territory)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O
(echo "== territory__rooms 8192 agents, beam skew 0.5"; bash tools/gpu_plan_sweep.sh "--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5" "3:3 3:4 3:2 4:4 2:3 2:2 3:5" "16"
echo "== commons 4096 agents"; bash tools/gpu_plan_sweep.sh "--substrate commons_harvest__open --obs agents" "3:6 3:3 3:4 3:5 4:4 2:3 4:6" "16"
echo "== pd arena 8192 agents"; bash tools/gpu_plan_sweep.sh "--substrate prisoners_dilemma_in_the_matrix__arena --obs agents --worlds 8192" "3:6 3:4 3:3 4:6 4:8 2:4" "16"
echo "== clean_up world"; bash tools/gpu_plan_sweep.sh "" "4:4 3:3 3:6 4:3 4:5 5:5" "12"; bash tools/gpu_plan_sweep.sh "" "4:4 3:3" "11 13 14"
) > $O/plans.txt 2>&1
cat $O/plans.txt
