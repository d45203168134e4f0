#!/bin/bash
# round 6, call 43: when do the workgroups of a headline launch end, by XCD (MP_FRAME_ENDS, tools/gpu_frame_ends.py) — call 42's
# four logged workgroups were done at 67 - 69 us of a launch of ~90: the stock ring, the team order, the pooled plans
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call43; mkdir -p $O
export MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_ends.so
for plan in "static_pct=100" "batch_worlds=1,ring_batches=8,team=1" "batch_worlds=1,ring_batches=8,static_pct=50" "batch_worlds=1,ring_batches=8,static_pct=75" "batch_worlds=4,ring_batches=2,static_pct=75"; do
  timeout 300 python tools/gpu_frame_ends.py clean_up 4096 world $plan 2>&1 | grep -v amdgpu.ids | tee -a $O/frame_ends.txt
done
