#!/bin/bash
# round 5, call 36: the whole GPU suite, smoke and the bench lines on the round's last library (call 24 again, after the plan rules of calls 29 and 34)
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r05_final3; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -9 $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-120
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --substrate territory__rooms --obs agents --players 9 --worlds 8192 --beam-skew 0.5 --steps 200 --warmup 300 --no-cpu-baseline --no-traffic > $O/bench_territory.json 2>> $O/bench.err; echo "territory rc=$?"
timeout 300 python bench.py --substrate commons_harvest__open --obs agents --players 16 --no-cpu-baseline --no-traffic > $O/bench_commons.json 2>> $O/bench.err; echo "commons rc=$?"
python - <<'PY'
import json
for f in ("bench", "bench_territory", "bench_commons"):
    l = json.loads(open(f"gpurun_out/r05_final3/{f}.json").read().strip().splitlines()[-1])
    print(f, round(l["value"] / 1e6, 1), round(l["ms_per_step"] * 1000, 1), round(l["roofline"]["frac"], 3),
          (round(l["substrate_api"]["value"] / 1e6, 1) if l.get("substrate_api") else None))
PY
