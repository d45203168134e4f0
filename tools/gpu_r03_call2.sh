#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03b; mkdir -p $O
for lib in "" batchwait; do
  echo "== lib [$lib]"
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_$lib.so
  MP_ENGINE_LIB=$L timeout 120 python tools/gpu_dbg_coins.py coins 120 3 2>&1 | grep -v amdgpu.ids
  MP_ENGINE_LIB=$L timeout 120 python tools/gpu_dbg_coins.py coins 120 0 2>&1 | grep -v amdgpu.ids | tail -8
done
timeout 120 python tools/gpu_dbg_coins.py clean_up 150 5 2>&1 | grep -v amdgpu.ids | tail -8
timeout 600 python -m pytest tests/test_gpu_surface.py tests/test_substrate_api.py -m gpu -q -x --timeout=300 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=300 -k "ring or natural" 2>&1 | tail -12
# renderer wave count sweep on the headline (waves = renderers + 4 feeders)
for wv in 8 9 10 11 12 14; do
  timeout 60 python bench.py --dev-plan waves=$wv --no-cpu-baseline --no-traffic --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves $wv frame %.1f us' % (d['kernels_ms']['frame']*1e3))"
done
