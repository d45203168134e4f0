#!/bin/bash
# round 4, call 5: range stealing (contiguous own range, then the fullest range's front)
set -u
out=gpurun_out/r04_fronts4; mkdir -p $out
b=tools/ubench/write_fronts
NBUF=5 VMM=2 STAMPS=1 VARIANTS=0 timeout 120 $b 0 1 > $out/steal_clean_up.md 2>&1; echo "rc $?"
NBUF=5 VMM=2 STAMPS=1 VARIANTS=0 timeout 120 $b 1 1 > $out/steal_commons.md 2>&1; echo "rc $?"
grep -v "| - | - |" $out/steal_clean_up.md; grep "then steal.*| - | - |" $out/steal_clean_up.md | head -40
grep -v "| - | - |" $out/steal_commons.md | head -30
