#!/bin/bash
# dev helper: print the frame plans (--dev-plan verbose=1) of the three bench configs
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1 MP_BENCH_ALLOW_DEV_ENV=1
for cfg in "" "--substrate commons_harvest__open --obs agents" "--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5"; do
  timeout -k 5 60 python -u bench.py --dev-plan verbose=1 --no-cpu-baseline --no-traffic --steps 5 --warmup 2 $cfg 2>&1 | grep "mp_engine:" | sort -u
done
