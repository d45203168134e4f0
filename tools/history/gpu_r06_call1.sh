#!/bin/bash
# round 6, call 1: the new tests (XCD-team dealing, mp_box_fill, eight ranks on one GPU, the ring's slot), then
# the team plan against the stock one on the same buffers — torch, mapped 2 MB, one contiguous extent —, the
# bench line with the driver's flags (configs legs + box_fill: how long does it take?), then the whole suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call1; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "xcd_teams or tuner_plans" --durations=5 ) > $O/pytest_team.log 2>&1; echo "team tests rc=$?"; tail -4 $O/pytest_team.log
( time timeout 600 python -m pytest tests/test_gpu_surface.py tests/test_gpu_ring.py -m gpu -x -q -k "box_fill or eight_engine or rollout_length or failed_placement or two_ranks" ) > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -4 $O/pytest_new.log
for cfg in "clean_up 4096 world" "clean_up 4096 both" "clean_up 4096 agents" "commons_harvest__open 4096 agents" "territory__rooms 8192 agents"; do
  set -- $cfg
  NBUF=3 MAPPED=3 CONTIG=1 timeout 400 python tools/gpu_paired_ab.py $1 $2 $3 - -:team_deal=2 -:batch_worlds=1,ring_batches=8 -:batch_worlds=1,ring_batches=8,team_deal=2 > $O/paired_$1_$3.txt 2>&1
  echo "== $cfg"; tail -12 $O/paired_$1_$3.txt | grep -v amdgpu.ids
done
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_flags.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r06_call1/bench_driver_flags.json").read().strip().splitlines()[-1])
print("headline", round(l["value"] / 1e6, 1), "M", round(l["roofline"]["avg_launch_ms"] * 1e3, 1), "us", round(l["roofline"]["frac"], 3), "plan", l["plan"])
print(" box_fill", json.dumps(l.get("box_fill")))
sa = l.get("substrate_api") or {}
print("substrate_api", round(sa.get("avg_launch_ms", 0) * 1e3, 1), "us", round(sa.get("frac", 0), 3), sa.get("plan"), json.dumps(sa.get("box_fill")))
for k, v in (l.get("configs") or {}).items():
  print(k, round(v["value"] / 1e6, 1), "M", round(v["avg_launch_ms"] * 1e3, 1), "us", round(v["frac"], 3), v["plan"], json.dumps(v["box_fill"]), "bind_s", v["placement"].get("bind_s"))
PY
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -12 $O/pytest_all.log
