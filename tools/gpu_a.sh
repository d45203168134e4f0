#!/bin/bash
# dev helper (GPU box): parity tests, then fused / unfused bench lines of the three BASELINE configs
cd $GRAFT_REPO_ROOT; O=gpurun_out/a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.1fM" % (d["value"]/1e6), "ms/step %.4f" % d["ms_per_step"], {k: round(v*1e3,1) for k,v in d["kernels_ms"].items()}, "frac", round(d["roofline"]["frac"],3))'
for rep in 1 2; do
for cfg in "" "--substrate commons_harvest__open --obs agents" "--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5"; do
  for mode in "" "--unfused"; do
    timeout 120 python bench.py --no-cpu-baseline --no-traffic --steps 100 $cfg $mode 2>$O/bench.err | tail -1 | python -c "$fmt" "${mode:-fused} ${cfg:0:30}" || tail -5 $O/bench.err
  done
done
done
