#!/bin/bash
# round 5, call 7: the placed ring (tests + bench rollout_api)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call7; mkdir -p $O
cd $R
( time timeout 600 python -m pytest tests/test_gpu_ring.py -m gpu -x -q --durations=8 ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -18 $O/pytest.log
( time timeout 600 python bench.py --no-cpu-baseline --no-traffic ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -3 $O/bench.err; python - <<'PY'
import json
l=json.loads(open("gpurun_out/r05_call7/bench.json").read().strip().splitlines()[-1])
print({k:l[k] for k in ("value","ms_per_step")}, l["roofline"]["frac"], l["placement"], l.get("steady_state"))
print("substrate_api", {k:l["substrate_api"][k] for k in ("value","ms_per_step","frac")})
r=l["rollout_api"]; print("rollout", r["single"], "\nclone", r["clone"], "\nring", r["ring"], r["ring_vs_single"], r["ring_vs_clone"])
PY
