#!/bin/bash
# round 6, call 33: territory's renderers wait 2 us a pass for their worlds (call 32): more feeders, now that the renderers
# are faster?  Same buffers, forced plans.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call33; mkdir -p $O
NBUF=2 MAPPED=4 timeout 900 python tools/gpu_paired_ab.py territory__rooms 8192 agents -:static_pct=100 -:feeders=6 -:batch_worlds=4,ring_batches=2,feeders=4 -:batch_worlds=2,ring_batches=4,feeders=4 -:batch_worlds=1,ring_batches=6,feeders=6 -:batch_worlds=1,ring_batches=6,feeders=6,team=1 -:batch_worlds=1,ring_batches=6,feeders=3,team=1 -:batch_worlds=1,ring_batches=8,feeders=4,team=1 > $O/territory_feeders.txt 2>&1; grep -v amdgpu.ids $O/territory_feeders.txt | tail -10
NBUF=2 MAPPED=4 timeout 900 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents -:static_pct=100 -:feeders=4,batch_worlds=4,ring_batches=2 -:batch_worlds=2,ring_batches=3,feeders=6 -:batch_worlds=1,ring_batches=6,feeders=6,team=1,late_feeder_prio=2 -:late_feeder_prio=2,pace=3 > $O/commons_feeders.txt 2>&1; grep -v amdgpu.ids $O/commons_feeders.txt | tail -10
