#!/bin/bash
# round 5, call 6: the whole GPU suite + smoke + the driver's bench flags + the default bench line
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call6; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -22 $O/pytest.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 $O/smoke.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
echo "bench(driver flags) rc=$?"; python - <<'PY'
import json
l=json.loads(open("gpurun_out/r05_call6/bench_driver_flags.json").read().strip().splitlines()[-1])
print({k:l[k] for k in ("value","ms_per_step")}, l["roofline"]["frac"], l["roofline"]["avg_launch_ms"], l["placement"], l.get("steady_state"))
print("substrate_api", {k:l["substrate_api"][k] for k in ("value","ms_per_step","frac")}, l["substrate_api"]["placement"])
r=l["rollout_api"]; print("rollout", r["single"]["value"], r["clone"]["value"], r["ring"]["value"], r["ring_vs_single"], r["ring"]["setup_s"])
PY
( time timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; python - <<'PY'
import json
l=json.loads(open("gpurun_out/r05_call6/bench.json").read().strip().splitlines()[-1])
print({k:l[k] for k in ("value","ms_per_step")}, l["roofline"]["frac"], l["roofline"]["avg_launch_ms"], l["placement"], l.get("steady_state"))
print("substrate_api", {k:l["substrate_api"][k] for k in ("value","ms_per_step","frac")})
r=l["rollout_api"]; print("rollout", r["single"]["value"], r["clone"]["value"], r["ring"]["value"], r["ring_vs_single"], r["ring"]["setup_s"])
PY
