#!/usr/bin/env python3
"""Does HOW a bound view was allocated decide how fast the frame launch writes it?

VERDICT r04 item 4: profiles/r04_write_fronts.md says "about 40 % of buffers are good, whatever
the method", profiles/r04_head.md section 3 saw 0 / 5 torch tensors fast against 3 / 3
engine-mapped ones — nobody counted.  This tool counts: K FRESH processes (a buffer's quality
is fixed for the life of its mapping, and a process's allocator hands the same physical pages
back) x {torch.empty, hipMalloc, chunks of 2 MB mapped, chunks of 32 MB mapped, one
physically contiguous extent} x {clean_up both views, commons_harvest per-agent}: the fused
launch under the STOCK plan (no tuner, no placement probe), 150 warm-up steps + 60 timed.

  python tools/alloc_method_study.py --processes 20 --out gpurun_out/r05_alloc   (GPU box)
  python tools/alloc_method_study.py --worker clean_up_both                       (one process)
"""
import argparse
import ctypes
import json
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

METHODS = ("torch.empty", "hipMalloc", "mapped 2 MB", "mapped 32 MB", "contiguous extent",
           "2 MB, 1 of 2", "2 MB, x4 shuffled", "64 KB shuffled")
CONFIGS = {"clean_up_both": ("clean_up", 4096, ("RGB", "WORLD.RGB")),
           "commons_agents": ("commons_harvest__open", 4096, ("RGB",)),
           "clean_up_world": ("clean_up", 4096, ("WORLD.RGB",))}


def worker(config, order_seed):
  import torch
  from meltingpot_amd import engine as E
  name, n, views = CONFIGS[config]
  eng = E.Engine(E.load_pack(name), n, placements=0)
  kinds = [E.OBS_RGB if v == "RGB" else E.OBS_WORLD_RGB for v in views]
  hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
  hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
  hip.hipFree.argtypes = [ctypes.c_void_p]
  L, dev = eng._L, eng.device.index or 0
  gen = torch.Generator(device=eng.device)
  gen.manual_seed(1)
  acts = torch.randint(0, eng.num_actions, (32, eng.N, eng.P), generator=gen, device=eng.device,
                       dtype=torch.int32)

  def nbytes(kind):
    shape, dtype = eng.shapes[kind]
    out = 1
    for d in shape:
      out *= d
    return out

  def alloc(method, kind):
    """-> (pointer, release())"""
    if method == "torch.empty":
      t = eng.empty(kind)
      return t.data_ptr(), (lambda: None), t
    p = ctypes.c_void_p()
    if method == "contiguous extent":
      rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes(kind), 0x4)
      if rc != 0 or not p.value:
        return None, None, None
      return p.value, (lambda: hip.hipFree(p)), None
    scattered = {"2 MB, 1 of 2": (2 << 20, 2, 0), "2 MB, x4 shuffled": (2 << 20, 4, 7 + order_seed),
                 "64 KB shuffled": (64 << 10, 1, 11 + order_seed)}
    if method in scattered:
      chunk, factor, seed = scattered[method]
      if L.mp_alloc_output_scattered(dev, nbytes(kind), chunk, factor, seed, ctypes.byref(p)) != 0:
        return None, None, None
      return p.value, (lambda: L.mp_free_output(dev, p)), None
    chunk = {"hipMalloc": 0, "mapped 2 MB": 2 << 20, "mapped 32 MB": 32 << 20}[method]
    if L.mp_alloc_output(dev, nbytes(kind), chunk, ctypes.byref(p)) != 0:
      return None, None, None
    return p.value, (lambda: L.mp_free_output(dev, p)), None

  def timed():
    for i in range(150):
      eng.step(acts[i % 32])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(60):
      eng.step(acts[i % 32])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 60 * 1e3

  order = list(METHODS)
  random.Random(order_seed).shuffle(order)
  eng.reset()
  out = {}
  for method in order:
    got = [alloc(method, k) for k in kinds]
    if any(g[0] is None for g in got):
      out[method] = None
    else:
      for k, g in zip(kinds, got):
        assert L.mp_bind_output(eng._h, k, ctypes.c_void_p(g[0])) == 0, L.mp_last_error()
      out[method] = round(timed(), 2)
      out[method + " again"] = round(timed(), 2)
    for k in kinds:
      L.mp_bind_output(eng._h, k, None)
    eng.sync()
    for g in got:
      if g[1]:
        g[1]()
  eng.close()
  print(json.dumps({"config": config, "order": order, "us": out}))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--worker")
  ap.add_argument("--seed", type=int, default=0)
  ap.add_argument("--processes", type=int, default=20)
  ap.add_argument("--configs", default="clean_up_both,commons_agents")
  ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_alloc"))
  args = ap.parse_args()
  if args.worker:
    return worker(args.worker, args.seed)
  os.makedirs(args.out, exist_ok=True)
  rows = []
  t0 = time.time()
  for i in range(args.processes):
    for config in args.configs.split(","):
      pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", config,
                           "--seed", str(i)], capture_output=True, text=True, timeout=300)
      lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
      if pr.returncode != 0 or not lines:
        rows.append({"config": config, "process": i, "error": (pr.stderr or "")[-300:]})
        continue
      rows.append(dict(json.loads(lines[-1]), process=i))
  with open(os.path.join(args.out, "alloc_method.json"), "w") as f:
    json.dump({"rows": rows, "seconds": round(time.time() - t0, 1),
               "host": os.uname().nodename}, f, indent=1)
  # the table: one row per process, one column per method; then the counts
  md = [f"alloc_method_study: {args.processes} fresh processes per config, stock plan, us per fused "
        f"launch (150 warm-up + 60 timed steps; second figure: the same buffer timed again), "
        f"{time.time() - t0:.0f} s\n"]
  for config in args.configs.split(","):
    mine = [r for r in rows if r["config"] == config and "us" in r]
    if not mine:
      continue
    best = min(v for r in mine for k, v in r["us"].items() if v)
    md.append(f"\n## {config}  (fastest launch seen: {best:.1f} us; 'fast' = within 6 % of it)\n")
    md.append("| process | " + " | ".join(METHODS) + " |")
    md.append("|---|" + "---:|" * len(METHODS))
    for r in mine:
      md.append(f"| {r['process']} | " + " | ".join(
          "-" if r["us"].get(m) is None else f"{r['us'][m]:.1f} / {r['us'][m + ' again']:.1f}"
          for m in METHODS) + " |")
    md.append("| **fast / measured** | " + " | ".join(
        f"{sum(1 for r in mine if r['us'].get(m) and r['us'][m] <= 1.06 * best)} / "
        f"{sum(1 for r in mine if r['us'].get(m))}" for m in METHODS) + " |")
    md.append("| median us | " + " | ".join(
        (lambda v: f"{sorted(v)[len(v) // 2]:.1f}" if v else "-")(
            [r["us"][m] for r in mine if r["us"].get(m)]) for m in METHODS) + " |")
  with open(os.path.join(args.out, "alloc_method.md"), "w") as f:
    f.write("\n".join(md) + "\n")
  print("\n".join(md))


if __name__ == "__main__":
  main()
