#!/bin/bash
# round 4, call 2: WHERE is the time lost on a slow buffer?  per-workgroup progress
# stamps; the same physical chunks mapped in three orders; busy work between spans
set -u
out=gpurun_out/r04_fronts2; mkdir -p $out
b=tools/ubench/write_fronts
STAMPS=1 VARIANTS=0,6,17 timeout 120 $b 0 1 > $out/stamps_clean_up.md 2>&1; echo "rc $?"
STAMPS=1 VARIANTS=0,6,17 timeout 120 $b 1 1 > $out/stamps_commons.md 2>&1; echo "rc $?"
NBUF=3 VMM=2,32,512 VARIANTS=0,6,9,17 timeout 120 $b 0 1 > $out/vmm_clean_up.md 2>&1; echo "rc $?"
NBUF=6 BUSY=600 VARIANTS=0,6,7,9,10,17 timeout 120 $b 0 1 > $out/busy600_clean_up.md 2>&1; echo "rc $?"
NBUF=6 BUSY=1200 VARIANTS=0,6,7,9,10,17 timeout 120 $b 0 1 > $out/busy1200_clean_up.md 2>&1; echo "rc $?"
cat $out/stamps_clean_up.md
