#!/bin/bash
# round 6, call 42: the head of the stepping launch on the final library, wave by wave (MP_FRAME_TIMELINE, tools/gpu_timeline.py):
# entry, barriers, requests, first data, record in LDS, hand-over
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call42; mkdir -p $O
export MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_timeline.so
UNTIL=22 timeout 200 python tools/gpu_timeline.py clean_up 4096 world 2>&1 | grep -v amdgpu.ids > $O/timeline_world.txt; grep -A13 "slot 0" $O/timeline_world.txt | cut -c1-330; grep -A13 "slot 2" $O/timeline_world.txt | cut -c1-330
