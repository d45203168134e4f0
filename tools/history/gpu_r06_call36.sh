#!/bin/bash
# round 6, call 36: territory's renderers wait for their worlds (call 32): six feeders at a lower priority than 3?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call36; mkdir -p $O
WARM=300 NBUF=2 MAPPED=4 timeout 900 python tools/gpu_paired_ab.py territory__rooms 8192 agents -:static_pct=100 -:late_feeder_prio=3 -:late_feeder_prio=2 -:feeders=6,late_feeder_prio=2 -:feeders=6,late_feeder_prio=3 -:feeders=6,late_feeder_prio=1 -:batch_worlds=1,ring_batches=6,feeders=3,team=1 -:batch_worlds=1,ring_batches=6,feeders=6,team=1,late_feeder_prio=2 > $O/territory_prio.txt 2>&1; grep -v amdgpu.ids $O/territory_prio.txt | tail -10
