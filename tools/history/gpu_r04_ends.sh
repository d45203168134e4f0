#!/bin/bash
set -u
out=gpurun_out/r04_ends; mkdir -p $out
export MP_ENGINE_LIB=$PWD/meltingpot_amd/lib/libmp_engine_ends.so PYTHONPATH=.
timeout 200 python tools/gpu_frame_ends.py clean_up 4096 world static_pct=100 > $out/clean_up_world.md 2>&1; echo "rc $?"
timeout 200 python tools/gpu_frame_ends.py clean_up 4096 world batch_worlds=1,ring_batches=8 > $out/clean_up_world_ring.md 2>&1; echo "rc $?"
timeout 200 python tools/gpu_frame_ends.py commons_harvest__open 4096 agents static_pct=100 > $out/commons.md 2>&1; echo "rc $?"
cat $out/*.md
