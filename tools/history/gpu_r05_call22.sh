#!/bin/bash
# round 5, call 22: first run of externality_mushrooms__dense on the GPU
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r05_mush; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_mushroom.py -q -m gpu --tb=short ) > $O/pytest1.log 2>&1
echo "rc=$?"; grep -v "^$" $O/pytest1.log | tail -70
