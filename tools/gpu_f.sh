#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/f; mkdir -p $O; export PYTHONUNBUFFERED=1
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.1fM" % (d["value"]/1e6), "ms/step %.4f" % d["ms_per_step"], {k: round(v*1e3,1) for k,v in d["kernels_ms"].items()}, "frac", round(d["roofline"]["frac"],3))'
for rep in 1 2; do
for cfg in "" "--substrate commons_harvest__open --obs agents" "--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5"; do
  for mode in "" "--unfused"; do
    timeout -k 5 60 python -u bench.py --no-cpu-baseline --no-traffic --steps 100 $cfg $mode 2>$O/bench.err | tail -1 | python -c "$fmt" "${mode:-fused} ${cfg:0:30}" || tail -3 $O/bench.err
  done
done
done
