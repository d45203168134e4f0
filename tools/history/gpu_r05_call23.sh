#!/bin/bash
# round 5, call 23: externality_mushrooms__dense in the shared suites (conformance, soak, fuzz,
# next orders, raw fields, full size) + its timings
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r05_mush; mkdir -p $O
( time timeout 1200 python -m pytest tests -q -m gpu -k "mushroom" --tb=short ) > $O/pytest2.log 2>&1
echo "rc=$?"; grep -v "^$" $O/pytest2.log | tail -40
timeout 300 python bench.py --substrate externality_mushrooms__dense --obs agents --steps 200 --warmup 20 --no-cpu-baseline --no-traffic > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
