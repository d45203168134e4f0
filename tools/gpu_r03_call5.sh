#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_matrix.py -m gpu -q -x --timeout=300 2>&1 | tail -6
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "frame %.1f us" % (d["kernels_ms"]["frame"]*1e3), "frac %.3f" % d["roofline"]["frac"])'
for rep in 1 2 3; do
for cfg in "" "--substrate commons_harvest__open --obs agents" "--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5" "--substrate prisoners_dilemma_in_the_matrix__arena --obs agents --worlds 8192 --fused"; do
  for dp in "no_helpers=1" "no_helpers=0"; do
    timeout -k 5 90 python -u bench.py --dev-plan $dp --no-cpu-baseline --no-traffic --steps 100 $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[$dp] ${cfg:12:30}"
  done
done; done
