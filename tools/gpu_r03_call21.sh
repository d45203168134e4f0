#!/bin/bash
# dev helper: same-box A/B of the renderer's instruction diet (base = HEAD's frame.hip) + parity of the paths it touches
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1
bash tools/gpu_ab.sh "base -" "" 2 > $O/ab.txt 2>&1; cat $O/ab.txt
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=240 --timeout-method=thread --durations=5 > $O/parity.log 2>&1
echo "parity rc=$? : $(tail -1 $O/parity.log)"
