#!/bin/bash
# round 6, call 8: the new resolve makes the launch its own store loop + head on every buffer (call 7: 94.7 us on the good
# one, 116 on three bad ones where the old, slower renderers gave 105 everywhere).  Does a bound on a wave's stores in
# flight (s_waitcnt vmcnt before the copy phase / before the pass / vmcnt(6)) give the bad buffers their 105 back?
# Fewer renderers?  Same buffers, paired.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call8; mkdir -p $O
NBUF=3 MAPPED=3 CONTIG=1 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 world - v1 w1 w2 w3 -:waves=10 -:waves=11 -:waves=14 > $O/paired_world.txt 2>&1; grep -v amdgpu.ids $O/paired_world.txt | tail -12
NBUF=3 MAPPED=3 CONTIG=1 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents - v1 w1 w2 w3 > $O/paired_agents.txt 2>&1; grep -v amdgpu.ids $O/paired_agents.txt | tail -12
NBUF=3 MAPPED=3 CONTIG=1 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents - v1 w1 w2 w3 > $O/paired_commons.txt 2>&1; grep -v amdgpu.ids $O/paired_commons.txt | tail -12
NBUF=3 MAPPED=3 CONTIG=1 timeout 600 python tools/gpu_paired_ab.py territory__rooms 8192 agents - v1 w1 w2 w3 > $O/paired_territory.txt 2>&1; grep -v amdgpu.ids $O/paired_territory.txt | tail -12
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 both - v1 w1 w2 > $O/paired_both.txt 2>&1; grep -v amdgpu.ids $O/paired_both.txt | tail -10
