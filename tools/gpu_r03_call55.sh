#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
fmt='import sys,json; d=json.loads(sys.stdin.read()); p=d["placement"]; print(sys.argv[1], "%.1f us" % (d["kernels_ms"]["frame"]*1e3), "frac %.3f" % d["roofline"]["frac"], "candidates", p["candidates"], "picked", p["kind"], "torch min %.0f mapped %s" % (min(p["dry_launch_us"][:12]), p["dry_launch_us"][12:18]))'
b() { timeout 100 python bench.py --no-cpu-baseline --no-traffic --steps 100 $2 2>/dev/null | tail -1 | python -c "$fmt" "$1"; }
(timeout -k 10 200 python -u -m pytest tests/test_gpu_surface.py -m gpu -q -x --timeout=240 --timeout-method=thread -k "placed or mapped or placing" 2>&1 | tail -1
for rep in 1 2 3; do b "headline"; done; b "commons" "--substrate commons_harvest__open --obs agents"; b "territory" "--substrate territory__rooms --obs agents --worlds 8192 --beam-skew 0.5 --warmup 300"; b "pd arena" "--substrate prisoners_dilemma_in_the_matrix__arena --obs agents --worlds 8192") > $O/place5.txt 2>&1
cat $O/place5.txt
