#!/usr/bin/env python3
"""dev helper (GPU box): the round-4 frame kernel — N-deep ring, pooled batches,
two views in one launch — against the oracle, every world, under several plans.
Each case reports instead of stopping the run."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import util
import test_gpu_parity as T
from meltingpot_amd import engine as E

CASES = [
    # (substrate, n, steps, fused, dev)
    ("clean_up", 64, 12, "world", None),
    ("clean_up", 64, 12, "world", {"batch_worlds": 1, "ring_batches": 8, "static_pct": 50, "max_groups": 4}),
    ("clean_up", 70, 12, "world", {"batch_worlds": 2, "ring_batches": 4, "static_pct": 50, "max_groups": 3}),
    ("clean_up", 300, 8, "world", {"batch_worlds": 1, "ring_batches": 8, "static_pct": 25, "max_groups": 7}),
    ("clean_up", 64, 12, "both", None),
    ("clean_up", 64, 12, "both", {"batch_worlds": 1, "ring_batches": 6, "static_pct": 50, "max_groups": 4}),
    ("clean_up", 64, 12, "agents", {"batch_worlds": 1, "ring_batches": 6, "static_pct": 50, "max_groups": 4}),
    ("commons_harvest__open", 40, 10, "agents", {"batch_worlds": 1, "ring_batches": 6, "static_pct": 50, "max_groups": 3}),
    ("commons_harvest__open", 40, 10, "both", None),
    ("territory__rooms", 40, 10, "agents", {"batch_worlds": 1, "ring_batches": 6, "feeders": 3, "static_pct": 50, "max_groups": 3}),
    ("territory__rooms", 40, 10, "both", {"static_pct": 60, "max_groups": 2}),
    ("coins", 130, 10, "both", {"static_pct": 50, "max_groups": 3}),
    ("prisoners_dilemma_in_the_matrix__arena", 40, 10, "both", {"static_pct": 50, "max_groups": 3}),
    ("clean_up", 1030, 5, "both", {"batch_worlds": 1, "ring_batches": 8, "static_pct": 50}),
    ("clean_up", 64, 8, "both", {"store_sc1": 1}),
    ("clean_up", 40, 6, "both", {"waves": 3, "feeders": 2, "batch_worlds": 2, "max_groups": 2}),
    ("clean_up", 40, 6, "both", {"waves": 2, "max_groups": 3}),
    ("commons_harvest__open", 40, 8, "agents", {"store_sc1": 1, "static_pct": 50, "max_groups": 3}),
]

def interleaved_launches():
  """Pooled step launches with draw-only launches (no pool) in between — the claim
  counters alternate by launch, whatever the launch's own plan."""
  pack = E.load_pack("clean_up")
  n = 96
  eng = E.Engine(pack, n, dev={"batch_worlds": 1, "ring_batches": 8, "static_pct": 50, "max_groups": 4})
  eng.bind(E.OBS_WORLD_RGB)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles: o.reset()
  rng = np.random.default_rng(3)
  acts = util.random_actions(rng, 9, n, eng.P, eng.num_actions)
  for s in range(9):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for k in range(s % 3):
      eng.observe(E.OBS_RGB if k else E.OBS_WORLD_RGB)   # draw-only launches
    for w, o in enumerate(oracles): o.step(acts[s, w])
    T._compare_state(eng, oracles, f"step {s + 1}")
  T._compare_rgb(eng, oracles, "end")
  eng.close()


# HEAD=<1 + mask>: every case under that FramePlan::head (how a stepping launch starts)
if os.environ.get("HEAD"):
  CASES = [(a, b, c, d, dict(e or {}, head=int(os.environ["HEAD"]))) for a, b, c, d, e in CASES]
bad = 0
try:
  interleaved_launches()
  print("OK   pooled steps interleaved with draw-only launches", flush=True)
except Exception as ex:  # pylint: disable=broad-except
  bad += 1
  print(f"FAIL interleaved launches: {type(ex).__name__}: {str(ex)[:400]}", flush=True)
for sub, n, steps, fused, dev in CASES:
  t0 = time.time()
  try:
    pack = E.load_pack(sub)
    T._run(pack, n=n, steps=steps, seed=n + steps, rgb_every=4, fused=fused, dev=dev,
           **({"unfused": False}))
    print(f"OK   {sub} n={n} fused={fused} dev={dev} ({time.time() - t0:.1f} s)", flush=True)
  except Exception as ex:  # pylint: disable=broad-except
    bad += 1
    print(f"FAIL {sub} n={n} fused={fused} dev={dev}: {type(ex).__name__}: {str(ex)[:400]}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
