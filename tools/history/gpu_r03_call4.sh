#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03d; mkdir -p $O
bash tools/gpu_tests.sh 300 900 2>&1 | tee $O/tests.txt
for t in tests/test_gpu_matrix.py; do grep -E "^E |FAILED|Error" gpurun_out/tests/test_gpu_matrix.log | head -20; done
echo "== matrix: product (12-wave) vs mx16 (16-wave, small spill)"
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "frame %.1f us" % (d["kernels_ms"]["frame"]*1e3), "frac %.3f" % d["roofline"]["frac"])'
for rep in 1 2; do for tag in - mx16; do
  lib=""; [ "$tag" != "-" ] && lib=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_$tag.so
  for cfg in "--substrate prisoners_dilemma_in_the_matrix__arena --obs agents --worlds 8192 --fused" "--substrate running_with_scissors_in_the_matrix__arena --obs world --worlds 4096 --fused"; do
    MP_BENCH_ALLOW_DEV_ENV=1 MP_ENGINE_LIB=$lib timeout -k 5 90 python -u bench.py --no-cpu-baseline --no-traffic --steps 100 $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$tag] ${cfg:12:50}"
  done
done; done
