#!/bin/bash
# round 6, call 7: the renderers' resolve rewritten (phase 1 v2: only the planes that can show anything, straight-line,
# selects on lane masks).  Paired on the same buffers: product (v2) | the old resolve (-DMP_P1_V1) | the old pass
# without the resolve (abl2) | without the resolve and its LDS reads (abl1) — then the GPU suite on the new library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call7; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_parity_64_worlds or finished or rgb" > $O/pytest_quick.log 2>&1; echo "quick parity rc=$?"; tail -3 $O/pytest_quick.log
NBUF=3 MAPPED=3 timeout 500 python tools/gpu_paired_ab.py clean_up 4096 world - v1 abl2 abl1 > $O/paired_world.txt 2>&1; grep -v amdgpu.ids $O/paired_world.txt | tail -11
NBUF=3 MAPPED=3 timeout 500 python tools/gpu_paired_ab.py clean_up 4096 agents - v1 abl2 abl1 > $O/paired_agents.txt 2>&1; grep -v amdgpu.ids $O/paired_agents.txt | tail -11
NBUF=3 MAPPED=3 timeout 500 python tools/gpu_paired_ab.py clean_up 4096 both - v1 > $O/paired_both.txt 2>&1; grep -v amdgpu.ids $O/paired_both.txt | tail -11
NBUF=3 MAPPED=3 timeout 500 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - v1 abl2 abl1 > $O/paired_commons.txt 2>&1; grep -v amdgpu.ids $O/paired_commons.txt | tail -11
NBUF=3 MAPPED=3 timeout 500 python tools/gpu_paired_ab.py territory__rooms 8192 agents - v1 abl2 abl1 > $O/paired_territory.txt 2>&1; grep -v amdgpu.ids $O/paired_territory.txt | tail -11
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 ) > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -12 $O/pytest_gpu.log
