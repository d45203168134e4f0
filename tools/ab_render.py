#!/usr/bin/env python3
"""In-process A/B of renderer variants (env overrides are read per launch), so
that box-to-box clock differences cancel: variants are interleaved round-robin
and the median per-launch time of each is printed.

  python tools/ab_render.py "MP_RENDER_WPB=8,MP_RENDER_WAVES=8" "MP_RENDER_STAGE=1,MP_RENDER_WPB=2,MP_RENDER_WAVES=4" ...
"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from meltingpot_amd import engine as E  # noqa: E402

KEYS = ("MP_RENDER_WPB", "MP_RENDER_WAVES", "MP_RENDER_STAGE", "MP_RENDER_ABLATE")


def main():
  obs_kind = E.OBS_RGB if os.environ.get("OBS") == "agents" else E.OBS_WORLD_RGB
  variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in sys.argv[1:]] or [{}]
  eng = E.Engine(E.load_pack(os.environ.get("SUBSTRATE", "clean_up")), 4096)
  eng.reset()
  acts = torch.randint(0, eng.num_actions, (32, eng.N, eng.P), device=eng.device, dtype=torch.int32)
  out = eng.empty(obs_kind)
  times = [[] for _ in variants]
  for rnd in range(24):
    eng.step(acts[rnd % 32])
    for i, v in enumerate(variants):
      for k in KEYS:
        os.environ.pop(k, None)
      os.environ.update(v)
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record(); eng.observe(obs_kind, out); b.record()
      torch.cuda.synchronize()
      if rnd >= 4:
        times[i].append(a.elapsed_time(b) * 1e3)
  for v, t in zip(variants, times):
    t.sort()
    print(f"{str(v):70s} median {t[len(t)//2]:7.1f} us  min {t[0]:7.1f}")


if __name__ == "__main__":
  main()
