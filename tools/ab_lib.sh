#!/bin/bash
# dev helper: same-box A/B of two engine builds (MP_ENGINE_LIB) on the bench configs
cd $GRAFT_REPO_ROOT
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.1fM" % (d["value"]/1e6), "render %.1f us" % (d["kernels_ms"]["render"]*1e3), "step %.1f us" % (d["kernels_ms"]["step"]*1e3), round(d["roofline"]["frac"],3))'
A=$GRAFT_REPO_ROOT/meltingpot_amd/lib/$1; shift
for rep in 1 2; do
  for lib in "$A" ""; do
    for cfg in "" "--substrate commons_harvest__open --obs agents" "--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5"; do
      MP_ENGINE_LIB=$lib timeout 100 python bench.py --no-cpu-baseline --steps 100 $cfg 2>&1 | tail -1 | python -c "$fmt" "${lib:+A}${lib:-B} ${cfg:0:24}"
    done
  done
done
