cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
fmt='import sys,json; d=json.loads(sys.stdin.read()); k=d["kernels_ms"]; print(sys.argv[1], "frame %.1f" % (k["frame"]*1e3), ("step %.1f render %.1f" % (k["step"]*1e3, k["render"]*1e3)) if "step" in k else "")'
for n in 256 512 1024 2048; do
  for m in --fused --unfused; do
  timeout -k 5 60 python -u bench.py --no-cpu-baseline --no-traffic --steps 200 --worlds $n $m 2>/dev/null | tail -1 | python -c "$fmt" "worlds $n $m"
  done
done
