#!/bin/bash
# round 4 (second session): the library the session started from ("old", 8318da0) against the
# committed one, the tuner on in both, same buffers; and the final build's first 20 us
set -u
out=gpurun_out/r04_final; mkdir -p $out
export PYTHONPATH=. NBUF=5 MAPPED=3
for sub in "clean_up 4096 world" "clean_up 4096 both" "commons_harvest__open 4096 agents" "territory__rooms 8192 agents"; do
  n=$(echo $sub | tr ' ' '_')
  timeout 300 python tools/gpu_paired_ab.py $sub old - old > $out/$n.txt 2>&1; echo "rc $?"
  grep -v amdgpu.ids $out/$n.txt
done
MP_ENGINE_LIB=$PWD/meltingpot_amd/lib/libmp_engine_timeline.so UNTIL=20 timeout 120 python tools/gpu_timeline.py clean_up 4096 world > $out/timeline.txt 2>&1
grep -v amdgpu.ids $out/timeline.txt | head -14 | grep "span\|wave  0\|wave  8\|wave  9"
