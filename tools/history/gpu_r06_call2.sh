#!/bin/bash
# round 6, call 2: how many concurrent streams should write a view?  The bare store loops over workgroups x
# storing waves on the same buffers (tools/ubench/fill_geometry.hip), then k_frame itself with more / fewer
# waves and workgroups on the same buffers (tools/gpu_paired_ab.py); the mushroom level's restated markings
# against the oracle (its own tests + the shared suites)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call2; mkdir -p $O
for s in 0 1 2; do timeout 300 tools/ubench/fill_geometry $s > $O/fill_geometry_$s.md 2>&1; echo "fill_geometry $s rc=$?"; done
head -12 $O/fill_geometry_0.md | cut -c1-400
NBUF=2 MAPPED=3 CONTIG=1 timeout 500 python tools/gpu_paired_ab.py clean_up 4096 world - -:waves=10 -:waves=14 -:waves=16 -:max_groups=228 -:max_groups=192 -:waves=10,max_groups=228 > $O/sweep_world.txt 2>&1; tail -12 $O/sweep_world.txt | grep -v amdgpu.ids
NBUF=2 MAPPED=3 CONTIG=1 timeout 500 python tools/gpu_paired_ab.py clean_up 4096 agents - -:feeders=3 -:waves=12 -:waves=14 -:feeders=3,waves=13 -:max_groups=192 > $O/sweep_agents.txt 2>&1; tail -12 $O/sweep_agents.txt | grep -v amdgpu.ids
NBUF=2 MAPPED=3 CONTIG=1 timeout 500 python tools/gpu_paired_ab.py clean_up 4096 both - -:feeders=3 -:feeders=3,world_waves=4 -:feeders=3,world_waves=8 -:waves=14,feeders=3 > $O/sweep_both.txt 2>&1; tail -12 $O/sweep_both.txt | grep -v amdgpu.ids
( time timeout 900 python -m pytest tests/test_gpu_mushroom.py tests/test_gpu_soak.py tests/test_every_substrate.py -m gpu -x -q -k "mushroom" --durations=5 ) > $O/pytest_mushroom.log 2>&1; echo "mushroom tests rc=$?"; tail -6 $O/pytest_mushroom.log
( time timeout 900 python tools/gpu_find_displaced_markings.py 16384 2500 0 ) > $O/find_displaced.txt 2>&1; echo "find rc=$?"; tail -15 $O/find_displaced.txt
