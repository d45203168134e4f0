#!/bin/bash
# dev helper: sweep render launch geometry.  usage: sweep.sh "<ablate list>" "wpb:waves ..."
for g in $2; do
  export MP_RENDER_WPB=${g%%:*} MP_RENDER_WAVES=${g##*:}
  for a in $1; do
    MP_RENDER_ABLATE=$a timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline ${OBS:+--obs $OBS} 2>&1 | tail -1 > /tmp/ab.json
    python -c "import json; d=json.load(open('/tmp/ab.json'))['kernels_ms']; print('wpb:waves=$g ablate=$a render %.1f (min %.1f)' % (d['render']*1e3, d['render_min']*1e3))"
  done
done
