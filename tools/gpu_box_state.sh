#!/bin/bash
# dev experiment: what do earlier processes do to a box that later launches run slower on it?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
fmt='import sys,json; d=json.loads(sys.stdin.read()); p=d["placement"]; print(sys.argv[1], "%.1f us" % (d["kernels_ms"]["frame"]*1e3), "probe min/median/max %.0f %.0f %.0f" % (min(p["dry_launch_us"]), sorted(p["dry_launch_us"])[6], max(p["dry_launch_us"])))'
b() { timeout 100 python bench.py --no-cpu-baseline --no-traffic --steps 100 $2 2>/dev/null | tail -1 | python -c "$fmt" "$1"; }
(b "fresh: headline"; b "fresh: commons" "--substrate commons_harvest__open --obs agents"
timeout 60 python - <<'PY'
import torch, random
random.seed(1)
live = []
for i in range(400):
  live.append(torch.empty(random.randint(1 << 20, 3 << 30), dtype=torch.uint8, device="cuda"))
  if len(live) > 12: del live[random.randrange(len(live))]
  if i % 50 == 0: torch.cuda.empty_cache()
print("churned: 400 allocations of 1 MB - 3 GB, no compute")
PY
b "after allocation churn: headline"; b "after allocation churn: commons" "--substrate commons_harvest__open --obs agents"
timeout 60 python - <<'PY'
import torch, time
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
t0 = time.time()
while time.time() - t0 < 15: x.add_(1)
torch.cuda.synchronize(); print("15 s of a busy GPU, one buffer")
PY
b "after 15 s busy: headline"; b "after 15 s busy: commons" "--substrate commons_harvest__open --obs agents"
timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=240 --timeout-method=thread 2>&1 | tail -1
b "after the parity tests: headline"; b "after the parity tests: commons" "--substrate commons_harvest__open --obs agents"
sleep 20
b "20 s later: headline"; b "20 s later: commons" "--substrate commons_harvest__open --obs agents") 2>&1 | grep -v amdgpu.ids > $O/box_state.txt
cat $O/box_state.txt
