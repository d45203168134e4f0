#!/bin/bash
# round 4 (second session): a frame's event log, old library against this one (timeline builds)
set -u
out=gpurun_out/r04_head; mkdir -p $out
export PYTHONPATH=. HEAD=1
for lib in oldtl timeline; do
  MP_ENGINE_LIB=$PWD/meltingpot_amd/lib/libmp_engine_$lib.so timeout 120 python tools/gpu_timeline.py clean_up 4096 ${1:-both} > $out/tl_${1:-both}_$lib.txt 2>&1; echo "$lib rc $?"
done
