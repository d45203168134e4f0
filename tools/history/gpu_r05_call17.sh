export TMPDIR=/tmp; O=gpurun_out/r05_call17; mkdir -p $O
for s in collaborative_cooking__cramped collaborative_cooking__crowded prisoners_dilemma_in_the_matrix__repeated coins; do
  for plan in "" "feeders=8" "feeders=12" "feeders=8,batch_worlds=16" "feeders=6,batch_worlds=6" "feeders=2"; do
    out=$(timeout 200 python bench.py --substrate $s --obs agents --no-cpu-baseline --no-substrate-api --no-rollout-api --no-steady-state --no-traffic ${plan:+--dev-plan $plan} 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print(round(l['ms_per_step']*1e3,1), l.get('plan'))")
    echo "$s [$plan] $out" | tee -a $O/sweep.txt
  done
done
