#!/bin/bash
# dev helper (GPU box): quick, individually time-limited checks of the new kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/b; mkdir -p $O; export PYTHONUNBUFFERED=1
run() { echo "=== $*"; timeout -k 5 "$@" 2>&1 | tail -${TAILN:-12}; echo "rc=${PIPESTATUS[0]}"; }
run 120 python -u -c "import __graft_entry__ as g; g.smoke()"
PT="python -u -m pytest -x -q --timeout=60 --timeout-method=thread -p no:cacheprovider"
run 100 $PT "tests/test_gpu_parity.py::test_reset_and_short_rollout[None]"
run 100 $PT "tests/test_gpu_parity.py::test_reset_and_short_rollout[agents]"
run 100 $PT "tests/test_gpu_parity.py::test_reset_and_short_rollout[world]"
run 100 $PT "tests/test_gpu_parity.py::test_reset_and_short_rollout[both]"
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.1fM" % (d["value"]/1e6), "ms/step %.4f" % d["ms_per_step"], {k: round(v*1e3,1) for k,v in d["kernels_ms"].items()}, "frac", round(d["roofline"]["frac"],3))'
for mode in "" "--unfused"; do
  echo "=== bench clean_up ${mode:-fused}"
  timeout -k 5 90 python -u bench.py --no-cpu-baseline --no-traffic --steps 60 $mode 2>$O/bench.err | tail -1 | python -c "$fmt" "${mode:-fused}" || tail -5 $O/bench.err
done
