#!/bin/bash
# round 4 (second session): where the kernel's arguments live — HIP_FORCE_DEV_KERNARG 0 / 1 / unset,
# alternating processes, same box (paired tool: old library, this one)
set -u
out=gpurun_out/r04_head; mkdir -p $out
export PYTHONPATH=. NBUF=6 MAPPED=0
for i in 1 2 3; do
  for kv in 0 1 unset; do
    if [ $kv = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$kv; fi
    echo "run $i HIP_FORCE_DEV_KERNARG=$kv: $(timeout 300 python tools/gpu_paired_ab.py clean_up 4096 world old:static_pct=100 -:static_pct=100 2>&1 | grep -v amdgpu.ids | tail -1)"
  done
done
