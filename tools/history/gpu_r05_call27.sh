#!/bin/bash
# round 5, call 27: plan sweep of the ninth level's fused launch (per-agent views), and of gift_refinements / coop_mining for comparison
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_sweep; mkdir -p $O
bash tools/gpu_plan_sweep.sh "--substrate externality_mushrooms__dense --obs agents" "2:2 2:3 3:3 3:4 4:4 4:6 6:6 2:4 1:4" "16 12" > $O/mush_agents.txt 2>&1
cat $O/mush_agents.txt
bash tools/gpu_plan_sweep.sh "--substrate externality_mushrooms__dense --obs world" "2:2 3:3 4:4 6:4 6:6 8:4 8:8" "16 12" > $O/mush_world.txt 2>&1
cat $O/mush_world.txt
