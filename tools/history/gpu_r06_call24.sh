#!/bin/bash
# round 6, call 24: where the pause does nothing the feeders are the pace (externality_mushrooms, call 23): their priority
# after the first world (late_feeder_prio = 1 + priority), on the levels whose plan has four feeders for sixteen worlds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call24; mkdir -p $O
for sub in externality_mushrooms__dense coop_mining gift_refinements; do
  NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py $sub 4096 agents -:static_pct=100 -:late_feeder_prio=2 -:late_feeder_prio=3 -:late_feeder_prio=4 -:late_feeder_prio=4,batch_worlds=1,ring_batches=8,team=1 -:late_feeder_prio=4,feeders=8,batch_worlds=4,ring_batches=2,waves=16 v1 > $O/prio_$sub.txt 2>&1; grep -v amdgpu.ids $O/prio_$sub.txt | tail -10
done
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py externality_mushrooms__dense 4096 world -:static_pct=100 -:late_feeder_prio=4 -:late_feeder_prio=4,pace=3 v1 > $O/prio_mushrooms_world.txt 2>&1; grep -v amdgpu.ids $O/prio_mushrooms_world.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py collaborative_cooking__crowded 4096 agents -:static_pct=100 -:late_feeder_prio=4 v1 > $O/prio_crowded.txt 2>&1; grep -v amdgpu.ids $O/prio_crowded.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 world -:static_pct=100 -:late_feeder_prio=2 -:late_feeder_prio=4 v1 > $O/prio_world.txt 2>&1; grep -v amdgpu.ids $O/prio_world.txt | tail -10
NBUF=2 MAPPED=4 timeout 600 python tools/gpu_paired_ab.py commons_harvest__open 4096 agents -:static_pct=100 -:late_feeder_prio=2 -:late_feeder_prio=4 v1 > $O/prio_commons.txt 2>&1; grep -v amdgpu.ids $O/prio_commons.txt | tail -10
