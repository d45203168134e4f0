#!/bin/bash
# round 4, call 4: per-XCD bandwidth to classes of a buffer's addresses
set -u
out=gpurun_out/r04_affinity; mkdir -p $out
timeout 120 tools/ubench/xcd_affinity 0 > $out/writes.md 2>&1; echo "rc $?"
timeout 120 tools/ubench/xcd_affinity 1 > $out/reads.md 2>&1; echo "rc $?"
head -120 $out/writes.md
