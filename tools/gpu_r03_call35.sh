#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
run() { # lib devplan substrate worlds
  [ -n "$1" ] && export MP_ENGINE_LIB=$R/meltingpot_amd/lib/libmp_engine_$1.so || unset MP_ENGINE_LIB
  echo "== lib [$1] plan [$2] $3: $(DEVPLAN=$2 timeout 120 python tools/gpu_bimodal3.py $3 $4 many_buffers 2>&1 | grep many | sed 's/.*step //; s/ us//' | tr '\n' ' ')"
}
for rep in 1 2; do
  run "" "" commons_harvest__open 4096
  run ntnone "" commons_harvest__open 4096
  run "" "feeders=3" commons_harvest__open 4096
  run "" "feeders=4,batch_worlds=4" commons_harvest__open 4096
  run "" "" territory__rooms 8192
  run ntnone "" territory__rooms 8192
done > $O/variants.txt 2>&1
cat $O/variants.txt
