#!/bin/bash
set -u
out=gpurun_out/r04_ring; mkdir -p $out
timeout 900 python tools/gpu_r04_ring_check.py > $out/ring_check.log 2>&1; echo "ring_check rc $?"
tail -30 $out/ring_check.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "geometry or ring_recycles or reset_and_short or render_paths" > $out/pytest_subset.log 2>&1; echo "pytest rc $?"
tail -5 $out/pytest_subset.log
