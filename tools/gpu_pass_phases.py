#!/usr/bin/env python3
"""dev helper (GPU box): where a renderer wave's pass goes.  Needs the -DMP_FRAME_TIMELINE build
(tools/ab_build.sh timeline -DMP_FRAME_TIMELINE) through MP_ENGINE_LIB.  Per renderer wave of the logged
workgroups, the mean microseconds between the stages of a pass: ticket taken (7) -> its worlds are
there (8) -> phase 1 done (20) -> composited cells staged (21) -> stores issued, pass done (9) -> next
ticket (7).  usage: gpu_pass_phases.py <substrate> <worlds> <world|agents|both> [k=v dev options]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meltingpot_amd import engine as E

name, worlds, vname = sys.argv[1], int(sys.argv[2]), sys.argv[3]
dev = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[4:])} or None
view = E.OBS_WORLD_RGB if vname == "world" else E.OBS_RGB
eng = E.Engine(E.load_pack(name), worlds, unfused=False, dev=dev, placements=0)
L = eng._L
L.mp_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
eng.bind(view)
if vname == "both":
  eng.bind(E.OBS_WORLD_RGB)
eng.reset()
acts = torch.randint(0, eng.num_actions, (40, worlds, eng.P), device=eng.device, dtype=torch.int32)
for s in range(30):
  eng.step(acts[s])
torch.cuda.synchronize()
n = 4 * 16 * 64 * 2
buf = np.zeros(n, np.uint32)
sums, counts = {}, {}
for rep in range(6):
  L.mp_debug_timeline(eng._h, buf.ctypes.data, n)   # clears the log
  eng.step(acts[30 + rep]); torch.cuda.synchronize()
  L.mp_debug_timeline(eng._h, buf.ctypes.data, n)
  log = buf.reshape(4, 16, 64, 2)
  for wg in range(4):
    for w in range(16):
      ev = [(int(c) & 255, int(t)) for c, t in log[wg, w] if c]
      if not any(c == 7 for c, _ in ev):
        continue                                     # a feeder
      for (c0, t0), (c1, t1) in zip(ev, ev[1:]):
        if c0 in (7, 8, 20, 21, 9) and c1 in (7, 8, 20, 21, 9):
          key = (c0, c1)
          sums[key] = sums.get(key, 0.0) + (t1 - t0) / 100.0
          counts[key] = counts.get(key, 0) + 1
names = {7: "ticket", 8: "worlds there", 20: "phase 1 done", 21: "staged", 9: "stores issued"}
print(f"{name} x{worlds} {vname} {dev or ''}: plan {eng.plan}")
total = 0.0
for key in [(7, 8), (8, 20), (20, 21), (21, 9), (9, 7)]:
  if key in counts:
    m = sums[key] / counts[key]
    total += m
    print(f"  {names[key[0]]:>14s} -> {names[key[1]]:<14s} {m:6.2f} us  (n = {counts[key]})")
print(f"  a pass: {total:.2f} us")
eng.close()
