#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1 TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=300 2>&1 | tail -2
fmt='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; print(sys.argv[1], {a: round(b*1e3,1) for a,b in k.items()})'
for rep in 1 2; do for tag in - nopf; do
  lib=""; [ "$tag" != "-" ] && lib=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_$tag.so
  for cfg in "" "--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5"; do
    MP_BENCH_ALLOW_DEV_ENV=1 MP_ENGINE_LIB=$lib timeout -k 5 120 python -u bench.py --cold --no-cpu-baseline --no-traffic --steps 100 $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$tag] ${cfg:12:20}"
  done
done; done
cd /tmp
for tag in - nopf; do
  lib=""; [ "$tag" != "-" ] && lib=$GRAFT_REPO_ROOT/meltingpot_amd/lib/libmp_engine_$tag.so
  MP_BENCH_ALLOW_DEV_ENV=1 MP_ENGINE_LIB=$lib timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $O/trace_$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $O/trace_$tag.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py --trace $O/trace_$tag/r_results.db --out $O/trace_$tag.md --title "trace [$tag]"; sed -n 5,8p $O/trace_$tag.md | cut -c1-200; tail -1 $O/trace_$tag.log | python -c "$fmt" "bench under rocprof [$tag]"
  rm -rf $O/trace_$tag
done
