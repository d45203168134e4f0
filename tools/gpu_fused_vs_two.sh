#!/bin/bash
# dev helper: fused vs two launches on the three bench configs, both forced (same box, one repetition)
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
fmt='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; print(sys.argv[1], "frame %.1f" % (k["frame"]*1e3), ("step %.1f render %.1f" % (k["step"]*1e3, k["render"]*1e3)) if "step" in k else "")'
for cfg in "" "--substrate commons_harvest__open --obs agents" "--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5"; do
  for mode in "--fused" "--unfused"; do
    timeout -k 5 60 python -u bench.py --no-cpu-baseline --no-traffic --steps 100 $cfg $mode 2>/dev/null | tail -1 | python -c "$fmt" "$mode ${cfg:12:22}"
  done
done
