#!/bin/bash
# round 6, call 31: where a renderer wave's cycles go — SQ wait / active counters by kind for the headline and for the per-agent
# launch (new library); the tuner after its margins were sorted (pause: 3 % dry or stepping; priority: stepping probes only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call31; mkdir -p $O
for cfg in "world:" "agents:--obs agents" "commons:--substrate commons_harvest__open --obs agents"; do
  name=${cfg%%:*}; args=${cfg#*:}
  PMC_SET="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" BENCH_ARGS="$args" bash tools/pmc_insts.sh gpurun_out/r06_call31/a_$name - 2>&1 | grep -v amdgpu.ids | tee -a $O/pmc_waits.txt | tail -10
  PMC_SET="SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" BENCH_ARGS="$args" bash tools/pmc_insts.sh gpurun_out/r06_call31/b_$name - 2>&1 | grep -v amdgpu.ids | tee -a $O/pmc_waits.txt | tail -10
done
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 world - -:static_pct=100 -:batch_worlds=1,ring_batches=8,team=1,pace=3 v1 > $O/tuned_world.txt 2>&1; grep -v amdgpu.ids $O/tuned_world.txt | tail -10
NBUF=3 MAPPED=3 timeout 600 python tools/gpu_paired_ab.py clean_up 4096 agents - -:static_pct=100 -:pace=5 v1 > $O/tuned_agents.txt 2>&1; grep -v amdgpu.ids $O/tuned_agents.txt | tail -10
