#!/bin/bash
# round 5, call 28: batches of 4 with 4 feeders against the stock per-agent plan (3 : 6, tuned to 3 : 3) on the levels with 5 - 7 viewers
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_sweep; mkdir -p $O
for cfg in "--substrate gift_refinements --obs agents" "--substrate coop_mining --obs agents" "--substrate clean_up --obs agents" "--substrate externality_mushrooms__dense --obs agents" "--substrate collaborative_cooking__crowded --obs agents"; do
  echo "== $cfg"
  bash tools/gpu_plan_sweep.sh "$cfg" "3:3 3:6 4:4 4:2 5:5" "16" 2>&1
done > $O/four_by_four.txt
cat $O/four_by_four.txt
