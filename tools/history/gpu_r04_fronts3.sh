#!/bin/bash
# round 4, call 3: dynamic balance (static contiguous share + chunks claimed from a
# device-wide counter) against the product order, slow and fast buffers side by side
set -u
out=gpurun_out/r04_fronts3; mkdir -p $out
b=tools/ubench/write_fronts
NBUF=4 VMM=2 STAMPS=1 VARIANTS=0 timeout 120 $b 0 1 > $out/dyn_clean_up.md 2>&1; echo "rc $?"
NBUF=4 VMM=2 STAMPS=1 VARIANTS=0 timeout 120 $b 1 1 > $out/dyn_commons.md 2>&1; echo "rc $?"
NBUF=4 VMM=2 VARIANTS=0 timeout 120 $b 0 0 > $out/dyn_clean_up_plain.md 2>&1; echo "rc $?"
cat $out/dyn_clean_up.md
