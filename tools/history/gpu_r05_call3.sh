#!/bin/bash
# round 5, call 3: address-bit locality by XCD subset; the rest of the GPU suite
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call3; mkdir -p $O
cd $R
( cd tools/ubench && timeout 300 ./class_bw 4 ) > $O/class_bw.md 2>&1
echo "class_bw rc=$?"; tail -22 $O/class_bw.md
( time timeout 600 python -m pytest tests/test_gpu_ring.py tests/test_gpu_surface.py tests/test_substrate_api.py tests/test_reference_wrappers.py tests/test_trace_replay.py -m gpu -x -q --durations=10 ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -30 $O/pytest.log
