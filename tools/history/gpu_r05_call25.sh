#!/bin/bash
# round 5, call 25: commons_harvest__open at 4096 x 16, this tree and the tree before the ninth level
# (a worktree under _ab_old/, not committed), alternating on one box
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_ab; mkdir -p $O
for k in 1 2 3; do
  for tree in new old; do
    d=$PWD; [ $tree = old ] && d=$PWD/_ab_old
    ( cd $d && PYTHONPATH=. timeout 300 python bench.py --substrate commons_harvest__open --obs agents --players 16 --no-cpu-baseline --no-traffic --no-steady-state > $O/commons_${tree}_$k.json 2>> $O/err.log )
    python - <<PY
import json
l = json.loads(open("$O/commons_${tree}_$k.json").read().strip().splitlines()[-1])
print("$tree", $k, round(l["ms_per_step"] * 1000, 1), sorted(l["placement"]["dry_launch_us"])[:3], sorted(l["placement"]["dry_launch_us"])[-1])
PY
  done
done
