"""dev helper: paired A/B of engine builds on the DRAW-ONLY launch (mp_observe: the render-only
k_frame, records loaded not stepped, plain stores) next to the fused step, same buffers —
what the renderers alone do with a buffer, without the feeders' steps beside them.

  python tools/gpu_draw_ab.py <substrate> <worlds> <world|agents> <libA> <libB> ...
  (lib as in tools/gpu_paired_ab.py)"""
import os, sys
import torch
from meltingpot_amd import engine as E

sub, n, view = sys.argv[1], int(sys.argv[2]), sys.argv[3]
tags = sys.argv[4:]
root = os.path.dirname(os.path.abspath(E.__file__))
kind = E.OBS_WORLD_RGB if view == "world" else E.OBS_RGB
pack = E.load_pack(sub)
engines = []
for spec in tags:
  tag, _, plan = spec.partition(":")
  dev = {k: int(v) for k, v in (kv.split("=") for kv in plan.split(",") if kv)}
  E._lib = None
  if tag == "-":
    os.environ.pop("MP_ENGINE_LIB", None)
  else:
    os.environ["MP_ENGINE_LIB"] = os.path.join(root, "lib", f"libmp_engine_{tag}.so")
  eng = E.Engine(pack, n, device=0, auto_reset=True, dev=dev or None, placements=0)
  eng.reset()
  engines.append(eng)
bufs = [engines[0].empty(kind) for _ in range(int(os.environ.get("NBUF", "3")))]
for _ in range(int(os.environ.get("MAPPED", "3"))):
  b = engines[0].empty_mapped(kind, 2 << 20)
  if b is not None:
    bufs.append(b)
gen = torch.Generator(device=engines[0].device); gen.manual_seed(5)
acts = torch.randint(0, engines[0].num_actions, (64, n, engines[0].P), generator=gen,
                     device=engines[0].device, dtype=torch.int32)
for eng in engines:
  for i in range(30): eng.step(acts[i % 64])


def timed(fn, reps=60):
  for _ in range(6): fn()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); a.record()
  for _ in range(reps): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / reps * 1e3


rows = []
for buf in bufs:
  row = []
  for eng in engines:
    eng.unbind(kind)
    row.append(timed(lambda: eng.observe(kind, out=buf)))
    eng.bind(kind, buf)
    k = [0]
    def step():
      eng.step(acts[k[0] % 64]); k[0] += 1
    row.append(timed(step))
    eng.unbind(kind)
  rows.append(row)
print(f"{sub} {view} x{n}: us per launch by buffer, per build [draw-only, fused step (stock plan)]: {tags}")
for row in rows: print("   " + "  ".join(f"{t:6.1f}" for t in row))
