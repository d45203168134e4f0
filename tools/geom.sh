#!/bin/bash
# dev helper: renderer launch-geometry sweep; usage: geom.sh "<bench args>" WPBxWAVES...
cd $GRAFT_REPO_ROOT
args=$1; shift
for g in "$@"; do
  export MP_RENDER_WPB=${g%x*} MP_RENDER_WAVES=${g#*x}
  timeout 100 python bench.py --no-cpu-baseline --steps 60 $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$g', 'render %.1f us' % (d['kernels_ms']['render']*1e3))" || echo "$g failed"
done
