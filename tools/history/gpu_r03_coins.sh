#!/bin/bash
# dev helper: the coins tests only (per-world maps and colours)
cd $GRAFT_REPO_ROOT; O=gpurun_out/coins; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py tests/test_substrate_api.py -m gpu -q -k "coins or raw_action or events_channel" --timeout=240 --timeout-method=thread --durations=5 > $O/coins.log 2>&1
echo "rc=$? : $(tail -1 $O/coins.log)"
grep -E "^(FAILED|ERROR)|Error" $O/coins.log | head -20
