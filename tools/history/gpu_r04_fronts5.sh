#!/bin/bash
# round 4: a static split calibrated on the previous launch's per-workgroup end times
set -u
out=gpurun_out/r04_fronts5; mkdir -p $out
b=tools/ubench/write_fronts
NBUF=5 VMM=2 VARIANTS=0 NO_DYN=1 NO_STEAL=1 CALIBRATE=1 timeout 120 $b 0 1 > $out/cal_xcd_clean_up.md 2>&1; echo "rc $?"
NBUF=5 VMM=2 VARIANTS=0 NO_DYN=1 NO_STEAL=1 CALIBRATE=2 timeout 120 $b 0 1 > $out/cal_wg_clean_up.md 2>&1; echo "rc $?"
NBUF=5 VMM=2 VARIANTS=0 NO_DYN=1 NO_STEAL=1 CALIBRATE=1 timeout 120 $b 1 1 > $out/cal_xcd_commons.md 2>&1; echo "rc $?"
cat $out/cal_xcd_clean_up.md $out/cal_wg_clean_up.md $out/cal_xcd_commons.md
