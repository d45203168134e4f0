#!/bin/bash
# dev helper: the whole `-m gpu` suite on the GPU box, file by file, each test
# bounded (pytest-timeout, thread method) and unbuffered so that a hang leaves a log.
cd $GRAFT_REPO_ROOT; O=gpurun_out/tests; mkdir -p $O; export PYTHONUNBUFFERED=1
for f in tests/test_gpu_matrix.py tests/test_gpu_surface.py tests/test_substrate_api.py tests/test_reference_wrappers.py tests/test_multi_gpu_sharding.py tests/test_gpu_parity.py; do
  n=$(basename $f .py)
  timeout -k 10 ${2:-900} python -u -m pytest $f -m gpu -q -x --timeout=${1:-240} --timeout-method=thread --durations=8 > $O/$n.log 2>&1
  echo "== $f rc=$? : $(tail -1 $O/$n.log)"
done
