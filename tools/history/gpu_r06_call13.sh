#!/bin/bash
# round 6, call 13: another phase between the workgroups' write fronts (workgroup g starts at its batch (g * rot) % owned):
# does any rotation serve the contiguous extent / the bad buffers as the good ones are served?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call13; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_parity_64_worlds or rgb" > $O/pytest_quick.log 2>&1; echo "quick parity rc=$?"; tail -2 $O/pytest_quick.log
NBUF=3 MAPPED=3 CONTIG=1 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 world -:static_pct=100 -:rotate=2 -:rotate=3 -:rotate=4 -:rotate=2,pace=3 -:pace=3 > $O/rot_world.txt 2>&1; grep -v amdgpu.ids $O/rot_world.txt | tail -12
NBUF=3 MAPPED=3 CONTIG=1 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents -:static_pct=100 -:rotate=2 -:rotate=3 -:rotate=4 -:rotate=6 > $O/rot_agents.txt 2>&1; grep -v amdgpu.ids $O/rot_agents.txt | tail -12
NBUF=3 MAPPED=3 CONTIG=1 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents -:static_pct=100 -:rotate=2 -:rotate=3 -:rotate=4 -:rotate=6 > $O/rot_commons.txt 2>&1; grep -v amdgpu.ids $O/rot_commons.txt | tail -12
