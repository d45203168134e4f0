#!/bin/bash
# dev helper: same-box comparison of several engine builds in meltingpot_amd/lib/ on one bench config
# usage: ab_libs.sh "<bench args>" lib1.so lib2.so ...   ("-" = the current build)
cd $GRAFT_REPO_ROOT
args=$1; shift
for rep in 1 2; do
  for l in "$@"; do
    lib=""; [ "$l" != "-" ] && lib=$GRAFT_REPO_ROOT/meltingpot_amd/lib/$l
    MP_ENGINE_LIB=$lib timeout 100 python bench.py --no-cpu-baseline --steps 100 $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', '%.1fM' % (d['value']/1e6), 'render %.1f us' % (d['kernels_ms']['render']*1e3))"
  done
done
