#!/bin/bash
# round 6, call 19: the copy phase with a test-free road for whole passes (six chunks bare, the others behind one scalar
# compare) and the per-pass divisions by host reciprocals: parity subset, then paired against the library before (pre)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call19; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "visible_planes or paced or geometry or both_views or short_rollout or 1000_fixed" --durations=5 ) > $O/pytest_sub.log 2>&1; echo "parity subset rc=$?"; tail -5 $O/pytest_sub.log
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 world - pre -:static_pct=100 pre:static_pct=100 > $O/paired_world.txt 2>&1; grep -v amdgpu.ids $O/paired_world.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents - pre -:static_pct=100 pre:static_pct=100 > $O/paired_agents.txt 2>&1; grep -v amdgpu.ids $O/paired_agents.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 both - pre -:static_pct=100 pre:static_pct=100 > $O/paired_both.txt 2>&1; grep -v amdgpu.ids $O/paired_both.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - pre -:static_pct=100 pre:static_pct=100 > $O/paired_commons.txt 2>&1; grep -v amdgpu.ids $O/paired_commons.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py territory__rooms 8192 agents - pre -:static_pct=100 pre:static_pct=100 > $O/paired_territory.txt 2>&1; grep -v amdgpu.ids $O/paired_territory.txt | tail -10
