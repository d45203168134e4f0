#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
for rep in 1 2; do
for lib in "" il; do
  [ -n "$lib" ] && export MP_ENGINE_LIB=$R/meltingpot_amd/lib/libmp_engine_$lib.so || unset MP_ENGINE_LIB
  echo "== lib [$lib]"
  timeout 120 python tools/gpu_bimodal3.py commons_harvest__open 4096 many_buffers 2>&1 | grep many | head -8 | cut -c1-60
done; done > $O/interleave.txt 2>&1
export MP_ENGINE_LIB=$R/meltingpot_amd/lib/libmp_engine_il.so
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=240 --timeout-method=thread > $O/parity_il.log 2>&1
echo "parity (interleaved) rc=$? : $(tail -1 $O/parity_il.log)" >> $O/interleave.txt
cat $O/interleave.txt
