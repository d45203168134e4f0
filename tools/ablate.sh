#!/bin/bash
# dev helper: render-kernel ablation (MP_RENDER_ABLATE bits: 1 no stores, 2 no compositing, 4 no state loads)
for a in "$@"; do
  MP_RENDER_ABLATE=$a timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline ${OBS:+--obs $OBS} 2>&1 | tail -1 > /tmp/ab.json
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('ablate=$a', d['kernels_ms'])"
done
