#!/bin/bash
# round 6, call 35: does the kernel — 17 K instructions with the resolve's twelve variants, 12.3 K before — wait for its
# instructions?  Instruction-cache and fetch counters of the headline launch and of the per-agent one (tools/pmc_icache.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call35; mkdir -p $O
bash tools/pmc_icache.sh --no-configs --no-box-fill 2>&1 | grep -v amdgpu.ids | tee $O/icache_world.txt | tail -16
rm -rf gpurun_out/pmc_icache
bash tools/pmc_icache.sh --no-configs --no-box-fill --obs agents 2>&1 | grep -v amdgpu.ids | tee $O/icache_agents.txt | tail -16
rm -rf gpurun_out/pmc_icache
