#!/bin/bash
# dev helper: per-phase cycle counts of one clean_up step (-DMP_STEP_TIMING build, stand-alone step kernel)
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
MP_ENGINE_LIB=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_steptiming.so timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu | tail -24
import torch
from meltingpot_amd import engine as E
eng = E.Engine(E.load_pack("clean_up"), 4096)
eng.reset()
acts = torch.randint(0, 9, (12, 4096, 7), device=eng.device, dtype=torch.int32)
for s in range(12):
  eng.step(acts[s])
torch.cuda.synchronize()
PY
