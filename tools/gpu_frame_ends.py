#!/usr/bin/env python3
"""dev helper (GPU box): when does every workgroup of a fused frame launch draw its
first pass and run out of passes?  Needs the -DMP_FRAME_ENDS build (tools/ab_build.sh
ends -DMP_FRAME_ENDS) loaded through MP_ENGINE_LIB.  Over several output buffers: the
launch time, the workgroups' end times (min / median / max) and their mean by XCD
(workgroup g runs on XCD g % 8) — profiles/r04_write_fronts.md section 2, in k_frame."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meltingpot_amd import engine as E

name = sys.argv[1] if len(sys.argv) > 1 else "clean_up"
worlds = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
view = E.OBS_WORLD_RGB if (len(sys.argv) <= 3 or sys.argv[3] == "world") else E.OBS_RGB
plan = {k: int(v) for k, v in (kv.split("=") for kv in (sys.argv[4] if len(sys.argv) > 4 else "static_pct=100").split(","))}
eng = E.Engine(E.load_pack(name), worlds, unfused=False, dev=plan, placements=0)
L = eng._L
L.mp_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
bufs = [eng.empty(view) for _ in range(5)] + [b for b in (eng.empty_mapped(view, 2 << 20) for _ in range(3)) if b is not None]
eng.reset()
acts = torch.randint(0, eng.num_actions, (16, worlds, eng.P), device=eng.device, dtype=torch.int32)
n = 2 * 256
print(f"{name} x{worlds} {'WORLD.RGB' if view == E.OBS_WORLD_RGB else 'RGB'}, plan {plan}: per buffer, us")
print("| buffer | launch | first pass: min / median / max | end: min / median / max | mean end by XCD 0..7 |")
print("|---|---:|---|---|---|")
for i, buf in enumerate(bufs):
  eng.bind(view, buf)
  for s in range(40):
    eng.step(acts[s % 16])
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for s in range(20):
    eng.step(acts[s % 16])
  b.record(); torch.cuda.synchronize()
  log = np.zeros(n, np.uint32)
  L.mp_debug_timeline(eng._h, log.ctypes.data, n)   # clears
  eng.step(acts[0]); torch.cuda.synchronize()
  L.mp_debug_timeline(eng._h, log.ctypes.data, n)
  log = log.reshape(256, 2).astype(np.int64)
  used = log[:, 1] != 0
  t0 = log[used, 0].min() if used.any() else 0
  first = (log[used, 0] - t0) / 100.0
  # (32-bit clock words: differences are what matters)
  end = ((log[used, 1] - t0) & 0xffffffff) / 100.0
  g = np.nonzero(used)[0]
  by_xcd = [end[g % 8 == x].mean() if (g % 8 == x).any() else 0 for x in range(8)]
  print(f"| {i}{' (mapped)' if i >= 5 else ''} | {a.elapsed_time(b) / 20 * 1e3:.1f} | {first.min():.1f} / {np.median(first):.1f} / {first.max():.1f} | "
        f"{end.min():.1f} / {np.median(end):.1f} / {end.max():.1f} | " + " ".join(f"{v:.0f}" for v in by_xcd) + " |", flush=True)
eng.close()
