#!/bin/bash
# dev helper: MP_RENDER_ABLATE sweep on arbitrary bench args; usage: ablate2.sh "<bench args>" v1 v2 ...
cd $GRAFT_REPO_ROOT
args=$1; shift
for a in "$@"; do
  MP_RENDER_ABLATE=$a timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$a render %.1f us' % (d['kernels_ms']['render']*1e3))"
done
