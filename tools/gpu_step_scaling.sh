#!/bin/bash
# dev helper: the stand-alone step kernels' time vs batch size (two launches per step, per-kernel events)
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
fmt='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; n=d["config"]["worlds_per_gpu"]; print(sys.argv[1], n, "worlds: step %.1f us (min %.1f), draw %.1f us, frame %.1f" % (k["step"]*1e3, k["step_min"]*1e3, k["render"]*1e3, k["frame"]*1e3))'
for cfg in "--substrate clean_up" "--substrate territory__rooms --obs agents --beam-skew 0.5" "--substrate commons_harvest__open --obs agents"; do
for n in 4096 16384 32768; do
  timeout -k 5 90 python -u bench.py --no-cpu-baseline --no-traffic --steps 40 --unfused --worlds $n $cfg 2>/dev/null | tail -1 | python -c "$fmt" "${cfg:12:20}"
done
done
