#!/bin/bash
# dev helper: the three bench configs, one short line each (value, kernel ms, roofline frac)
cd $GRAFT_REPO_ROOT
fmt='import sys,json; d=json.loads(sys.stdin.read()); print("%.1fM" % (d["value"]/1e6), {k: round(v,4) for k,v in d["kernels_ms"].items()}, round(d["roofline"]["frac"],3))'
for i in 1 2; do timeout 100 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "$fmt"; done
timeout 100 python bench.py --no-cpu-baseline --substrate commons_harvest__open --obs agents "$@" 2>&1 | tail -1 | python -c "$fmt"
timeout 100 python bench.py --no-cpu-baseline --substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5 "$@" 2>&1 | tail -1 | python -c "$fmt"
