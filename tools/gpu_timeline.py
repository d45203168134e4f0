#!/usr/bin/env python3
"""dev helper (GPU box): where the time of one frame goes.  Needs the
-DMP_FRAME_TIMELINE build (tools/ab_build.sh timeline -DMP_FRAME_TIMELINE) loaded
through MP_ENGINE_LIB; prints, for workgroups 0, 1, 128 and the last one, every
wave's pipeline events of ONE step (stage, value, microseconds since the
workgroup's first event).
  stages: 1 entry, 2 prologue copied, 3 past the barrier; feeders: 4 batch k (before the
  buffer is free), 5 slot (buffer free, before the load), 6 slot published, 15 exit;
  11 (round 4) the feeder's first data is there and the feeders have met, 12 record in LDS;
  renderers: 7 ticket taken, 8 its worlds are there, 9 pass done, 14 exit."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meltingpot_amd import engine as E

name = sys.argv[1] if len(sys.argv) > 1 else "clean_up"
worlds = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
vname = sys.argv[3] if len(sys.argv) > 3 else "world"   # world | agents | both
view = E.OBS_WORLD_RGB if vname == "world" else E.OBS_RGB
# HEAD=<1 + mask>: FramePlan::head (how a stepping launch starts); UNTIL=<us>: only the events before
dev = {"head": int(os.environ["HEAD"])} if os.environ.get("HEAD") else None
until = float(os.environ.get("UNTIL", "1e9"))
eng = E.Engine(E.load_pack(name), worlds, unfused=False, dev=dev)
L = eng._L
L.mp_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
eng.bind(view)
if vname == "both":
  eng.bind(E.OBS_WORLD_RGB)
eng.reset()
acts = torch.randint(0, eng.num_actions, (8, worlds, eng.P), device=eng.device, dtype=torch.int32)
for s in range(6):
  eng.step(acts[s])
torch.cuda.synchronize()
n = 4 * 16 * 64 * 2
buf = np.zeros(n, np.uint32)
L.mp_debug_timeline(eng._h, buf.ctypes.data, n)   # clears the log
eng.step(acts[6]); torch.cuda.synchronize()
L.mp_debug_timeline(eng._h, buf.ctypes.data, n)
log = buf.reshape(4, 16, 64, 2)
for wg in range(4):
  ev = [(int(t), w, int(c) & 255, int(c) >> 8) for w in range(16) for c, t in log[wg, w] if c]
  if not ev: continue
  t0 = min(e[0] for e in ev)
  print(f"--- workgroup slot {wg}: {len(ev)} events, span {(max(e[0] for e in ev) - t0) / 100:.1f} us")
  for w in range(16):
    row = [(t, c, v) for t, ww, c, v in ev if ww == w]
    if row:
      print(f"  wave {w:2d}: " + " ".join(f"{c}:{v}@{(t - t0) / 100:.1f}" for t, c, v in row
                                        if (t - t0) / 100 <= until))
eng.close()
