#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1; O=gpurun_out/r03n; mkdir -p $O
bash tools/gpu_tests.sh 300 900 2>&1 | tee $O/tests.txt
grep -E "^E |FAILED" gpurun_out/tests/*.log | head -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
