#!/bin/bash
# round 6, call 4: the round's profile set (kernel traces + HBM traffic passes of the three bench configs and of
# the substrate_api leg) on the round's library
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call4; mkdir -p $O
cd $R
timeout 1700 bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; echo "profile rc=$?"; tail -5 $O/profile_round.log
ls gpurun_out/prof_r06/
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_r06/*.bench.json")):
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(l["value"]/1e6,1), round(l["ms_per_step"]*1e3,2), round(l["roofline"]["avg_launch_ms"]*1e3,2), round(l["roofline"]["frac"],3), l["roofline"]["traffic"])
    except Exception as e:
        print(f, "unreadable", e)
PY
