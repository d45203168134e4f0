#!/bin/bash
# round 4, call 1: which span ORDER makes a persistent writer placement-insensitive?
# (tools/ubench/write_fronts.hip; profiles/r04_write_fronts.md)
set -u
out=gpurun_out/r04_fronts; mkdir -p $out
b=tools/ubench/write_fronts
[ -x $b ] || hipcc --offload-arch=gfx950 -O3 -o $b $b.hip
for run in "0 1" "1 1" "0 0" "0 2" "2 1" "3 1"; do
  set -- $run
  timeout 120 $b $1 $2 > $out/shape$1_policy$2.md 2>&1
  echo "shape $1 policy $2 rc $?"
done
# a second process: are the per-buffer numbers a property of the process's buffers?
timeout 120 $b 0 1 > $out/shape0_policy1_again.md 2>&1
cat $out/shape0_policy1.md
