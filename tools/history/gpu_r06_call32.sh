#!/bin/bash
# round 6, call 32: where a renderer wave's pass goes (MP_FRAME_TIMELINE build, tools/gpu_pass_phases.py): the headline, the
# per-agent view, both views, commons_harvest
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call32; mkdir -p $O
export MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_timeline.so
for cfg in "clean_up 4096 world" "clean_up 4096 world batch_worlds=1 ring_batches=8 team=1" "clean_up 4096 agents" "clean_up 4096 agents feeders=3" "clean_up 4096 agents pace=5" "clean_up 4096 both" "commons_harvest__open 4096 agents" "territory__rooms 8192 agents"; do
  timeout 200 python tools/gpu_pass_phases.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $O/pass_phases.txt
done
