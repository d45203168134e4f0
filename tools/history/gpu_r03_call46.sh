#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
fmt='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "%.1f us" % (d["kernels_ms"]["frame"]*1e3), "frac %.3f" % d["roofline"]["frac"], d.get("placement"), [round(x*1e3) for x in d["kernels_ms"].get("frame_by_placement", [])])'
for rep in 1 2; do
for cfg in "" "--substrate commons_harvest__open --obs agents" "--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5 --warmup 300"; do
  for place in 8 1; do
    timeout -k 5 100 python -u bench.py --no-cpu-baseline --no-traffic --steps 100 --place $place $cfg 2>/dev/null | tail -1 | python -c "$fmt" "[place $place] ${cfg:12:18}"
  done
done; done > $O/place2.txt 2>&1
timeout 100 python -u bench.py --no-cpu-baseline --no-traffic --steps 100 --place 16 --placements 6 2>/dev/null | tail -1 | python -c "$fmt" "[place 16 + sweep]" >> $O/place2.txt
cat $O/place2.txt
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_substrate_api.py -m gpu -q -x --timeout=240 --timeout-method=thread 2>&1 | tail -2
