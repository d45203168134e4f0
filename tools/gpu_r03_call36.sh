#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
run() { # lib devplan substrate worlds
  [ -n "$1" ] && export MP_ENGINE_LIB=$R/meltingpot_amd/lib/libmp_engine_$1.so || unset MP_ENGINE_LIB
  echo "== lib [$1] plan [$2] $3: $(DEVPLAN=$2 timeout 120 python tools/gpu_bimodal3.py $3 $4 many_buffers 2>&1 | grep many | sed 's/.*step //; s/ us//' | tr '\n' ' ')"
}
for rep in 1 2; do
  for lib in "" tok2 tok3 tok4 tok6; do run "$lib" "" commons_harvest__open 4096; done
done > $O/tokens.txt 2>&1
cat $O/tokens.txt
