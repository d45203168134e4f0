export TMPDIR=/tmp PYTHONPATH=.
timeout 900 python -m pytest tests/test_gpu_cook.py tests/test_gpu_matrix.py tests/test_gpu_soak.py tests/test_every_substrate.py -m gpu -x -q 2>&1 | tail -3
for s in collaborative_cooking__cramped prisoners_dilemma_in_the_matrix__repeated coins collaborative_cooking__crowded; do
  for v in world both; do
    NBUF=1 MAPPED=3 timeout 300 python tools/gpu_paired_ab.py $s 4096 $v - 2>&1 | grep -E "mean" | sed "s/^/$s $v /"
  done
done
