"""round 5: in a process whose allocator state makes plain mapped-2-MB views mostly SLOW (behind a
placement probe's 16 GB of churn, like bench.py's rollout leg), are views built from a seeded
pick out of a larger pool of chunks reliably fast?  clean_up x 4096, per-agent RGB, stock plan."""
import ctypes, sys, time
import torch
from meltingpot_amd import engine as E

eng = E.Engine(E.load_pack("clean_up"), 4096)
L, dev = eng._L, 0
kind = E.OBS_RGB
nbytes = eng.observe(E.OBS_REWARD).numel() * 0 + 4096 * 7 * 88 * 88 * 3
t0 = time.time()
first = eng.bind(kind)            # the product's placement: up to 24 candidates
print("placement:", eng.placement[kind]["dry_launch_us"], f"{time.time() - t0:.1f} s")
eng.unbind(kind); del first
eng.placements = 0
gen = torch.Generator(device=eng.device); gen.manual_seed(1)
acts = torch.randint(0, eng.num_actions, (32, eng.N, eng.P), generator=gen, device=eng.device, dtype=torch.int32)
eng.reset()

def timed(ptr):
  assert L.mp_bind_output(eng._h, kind, ctypes.c_void_p(ptr)) == 0
  for i in range(100): eng.step(acts[i % 32])
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for i in range(60): eng.step(acts[i % 32])
  b.record(); torch.cuda.synchronize()
  L.mp_bind_output(eng._h, kind, None)
  return a.elapsed_time(b) / 60 * 1e3

methods = {"plain": (1, 0), "1 of 2": (2, 0), "x2 shuffled": (2, 1), "x4 shuffled": (4, 1), "x8 shuffled": (8, 1),
           "x16 shuffled": (16, 1)}
rows = {m: [] for m in methods}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
  for m, (factor, seeded) in methods.items():
    p = ctypes.c_void_p()
    t0 = time.time()
    rc = L.mp_alloc_output_scattered(dev, nbytes, 2 << 20, factor, (rep * 16 + factor) if seeded else 0, ctypes.byref(p))
    if rc != 0:
      rows[m].append(None); continue
    alloc_s = time.time() - t0
    rows[m].append((round(timed(p.value), 1), round(alloc_s, 2)))
    L.mp_free_output(dev, p)
for m, v in rows.items():
  print(f"{m:14s}", v)
