#!/bin/bash
# round 6, call 37: the other levels' bench lines on the final library (the round-5 figures are the old resolve's):
# -> gpurun_out/r06_call37/other_levels.json
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call37; mkdir -p $O
python - <<'PY'
import json, subprocess, sys
out = {}
cfgs = [("coop_mining", "agents"), ("coop_mining", "world"), ("gift_refinements", "agents"), ("gift_refinements", "world"),
        ("externality_mushrooms__dense", "agents"), ("externality_mushrooms__dense", "world"),
        ("collaborative_cooking__crowded", "agents"), ("collaborative_cooking__crowded", "world"),
        ("collaborative_cooking__cramped", "agents"), ("collaborative_cooking__cramped", "world"),
        ("coins", "agents"), ("prisoners_dilemma_in_the_matrix__arena", "agents"),
        ("prisoners_dilemma_in_the_matrix__repeated", "agents"), ("territory__open", "agents"),
        ("commons_harvest__closed", "agents"), ("clean_up", "agents")]
for sub, obs in cfgs:
  r = subprocess.run([sys.executable, "bench.py", "--substrate", sub, "--obs", obs, "--steps", "200", "--warmup", "100",
                      "--no-cpu-baseline", "--no-traffic", "--no-substrate-api", "--no-rollout-api", "--no-steady-state",
                      "--no-configs"], capture_output=True, text=True, timeout=300)
  try:
    l = json.loads(r.stdout.strip().splitlines()[-1])
  except Exception as e:
    print(sub, obs, "FAILED", r.stderr[-300:]); continue
  out[f"{sub}_{obs}"] = l
  p = {k: v for k, v in l["plan"].items() if k in ("batch_worlds", "feeders", "pace", "xcd_teams", "late_feeder_priority", "sc1_stores")}
  print(f"{sub:45s} {obs:6s} {l['value'] / 1e6:7.1f} M  {l['roofline']['avg_launch_ms'] * 1e3:6.1f} us  {l['roofline']['frac']:.3f}  of box fill {l['box_fill']['frac_of_box_fill']:.2f}  {p}", flush=True)
json.dump(out, open("gpurun_out/r06_call37/other_levels.json", "w"), indent=1)
PY
