#!/bin/bash
# round 4: which physical chunk size makes a mapped view fast?
set -u
out=gpurun_out/r04_fronts6; mkdir -p $out
b=tools/ubench/write_fronts
for i in 1 2 3; do
NBUF=2 NO_CONTIG=1 VMM_LAYOUTS=1 VMM=64k,256k,1,2,4,8,16,32 VARIANTS=0 NO_DYN=1 NO_STEAL=1 timeout 120 $b 0 1 > $out/chunks_clean_up_$i.md 2>&1; echo "rc $?"
done
NBUF=2 NO_CONTIG=1 VMM_LAYOUTS=1 VMM=64k,256k,1,2,4,8,16,32 VARIANTS=0 NO_DYN=1 NO_STEAL=1 timeout 120 $b 1 1 > $out/chunks_commons.md 2>&1; echo "rc $?"
cat $out/*.md
