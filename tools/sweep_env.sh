#!/bin/bash
# dev helper: sweep one environment variable over values on a bench config; usage: sweep_env.sh VAR "<bench args>" v1 v2 ...
cd $GRAFT_REPO_ROOT
var=$1; args=$2; shift 2
for v in "$@"; do
  export $var=$v
  MP_RENDER_VERBOSE=1 timeout 100 python bench.py --no-cpu-baseline --steps 60 $args 2>/tmp/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', 'render %.1f us' % (d['kernels_ms']['render']*1e3), '%.1fM' % (d['value']/1e6))" || echo "$v failed"
  grep "mp_engine" /tmp/err.txt | tr '\n' ' '; echo
done
