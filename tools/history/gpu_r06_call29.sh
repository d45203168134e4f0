#!/bin/bash
# round 6, call 29: a per-agent pass with the viewer's head bytes in one LDS round trip and its OutOfBounds image from a
# per-viewer table (library "base" = before): parity subset, then paired, forced stock plans
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call29; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_matrix.py -m gpu -x -q -k "visible_planes or geometry or both_views or short_rollout or torus or arena or repeated" --durations=5 ) > $O/pytest_sub.log 2>&1; echo "parity subset rc=$?"; tail -4 $O/pytest_sub.log
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents -:static_pct=100 base:static_pct=100 -:feeders=3 base:feeders=3 > $O/oob_agents.txt 2>&1; grep -v amdgpu.ids $O/oob_agents.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents -:static_pct=100 base:static_pct=100 -:batch_worlds=1,ring_batches=6,team=1 base:batch_worlds=1,ring_batches=6,team=1 > $O/oob_commons.txt 2>&1; grep -v amdgpu.ids $O/oob_commons.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py territory__rooms 8192 agents -:static_pct=100 base:static_pct=100 > $O/oob_territory.txt 2>&1; grep -v amdgpu.ids $O/oob_territory.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 both -:static_pct=100 base:static_pct=100 -:feeders=3 base:feeders=3 > $O/oob_both.txt 2>&1; grep -v amdgpu.ids $O/oob_both.txt | tail -10
