#!/bin/bash
# round 6, call 6: the one world the deep soak found different (coins, world 835 of 2048): against the oracle after
# every step — in the batch as the soak ran it, with WORLD.RGB instead, with no view, and alone (16 worlds at offset 832)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call6; mkdir -p $O
for v in "2048 900 agents" "2048 900 world" "2048 900 none" "2048 900 agents alone"; do
  echo "== coins 835 $v"; timeout 600 python tools/gpu_trace_world.py coins 835 $v 2>&1 | grep -v amdgpu.ids | tee -a $O/trace_coins_835.txt | tail -12
done
