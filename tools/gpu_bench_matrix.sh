#!/bin/bash
# dev helper: bench lines of the *_in_the_matrix level (not a BASELINE.json config)
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.1fM agent-steps/s" % (d["value"]/1e6), "ms/step %.4f" % d["ms_per_step"], {k: round(v*1e3,1) for k,v in d["kernels_ms"].items()}, "frac", round(d["roofline"]["frac"],3), "players", d["config"]["players"])'
for cfg in "--substrate prisoners_dilemma_in_the_matrix__arena --obs agents --worlds 8192" "--substrate prisoners_dilemma_in_the_matrix__arena --obs agents --worlds 8192 --unfused" "--substrate prisoners_dilemma_in_the_matrix__repeated --obs agents --worlds 16384" "--substrate prisoners_dilemma_in_the_matrix__repeated --obs agents --worlds 16384 --unfused" "--substrate running_with_scissors_in_the_matrix__arena --obs world --worlds 4096"; do
  timeout -k 5 90 python -u bench.py --no-cpu-baseline --no-traffic --steps 100 $cfg 2>gpurun_out/mx.err | tail -1 | python -c "$fmt" "${cfg:12:60}" || tail -3 gpurun_out/mx.err
done
