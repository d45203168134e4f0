#!/usr/bin/env python3
"""Record a DMLab2D trace to fit the oracle's engine assumptions (DESIGN.md §5,
A1-A9).  Needs the `dmlab2d` wheel + the reference's Python deps: NOT runnable
in the build container; ships for whoever has them.

  python tools/dump_dmlab2d_trace.py --out trace_clean_up.npz
         [--substrate clean_up] [--players 7] [--steps 1000] [--seed 1]

Writes actions [T, P], rewards [T, P], WORLD.RGB [T+1, H, W, 3], per-player RGB
[T+1, P, 88, 88, 3] and (with _ENABLE_DEBUG_OBSERVATIONS patched on) POSITION /
ORIENTATION.  `tests/tools/replay_trace.py` is the other half: it replays the file
through the oracle under every combination of the assumption switches and
reports where each first diverges.  Per-draw RNG
values can never match (A10: Philox vs mt19937_64), so only draws-free
behaviour — movement, blocking, beam footprints, view rotation, compositing,
sprite down-scaling — is fitted from traces of deterministic situations.
"""
import argparse

import numpy as np


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--substrate", default="clean_up")
  ap.add_argument("--players", type=int, default=7)
  ap.add_argument("--steps", type=int, default=1000)
  ap.add_argument("--seed", type=int, default=1)
  ap.add_argument("--out", required=True)
  a = ap.parse_args()

  from meltingpot import substrate  # the reference package
  from meltingpot.utils.substrates import builder
  import importlib
  cfg_mod = importlib.import_module(f"meltingpot.configs.substrates.{a.substrate}")
  config = substrate.get_config(a.substrate)
  roles = ("default",) * a.players
  settings = cfg_mod.build(roles, config)
  env = builder.builder(settings, env_seed=a.seed)  # raw dmlab2d env, "N.KEY" dicts
  rng = np.random.default_rng(a.seed)
  n_actions = len(config.action_set)
  ts = env.reset()
  world, rgb, acts, rews, pos, ori = [], [], [], [], [], []

  def snap(t):
    world.append(np.array(t.observation["WORLD.RGB"]))
    rgb.append(np.stack([t.observation[f"{p + 1}.RGB"] for p in range(a.players)]))
    if "1.POSITION" in t.observation:   # _ENABLE_DEBUG_OBSERVATIONS (teacher forcing)
      pos.append(np.stack([t.observation[f"{p + 1}.POSITION"] for p in range(a.players)]))
      ori.append(np.stack([t.observation[f"{p + 1}.ORIENTATION"] for p in range(a.players)]))

  snap(ts)
  for _ in range(a.steps):
    ids = rng.integers(0, n_actions, a.players)
    flat = {}
    for p, i in enumerate(ids):
      for k, v in config.action_set[i].items():
        flat[f"{p + 1}.{k}"] = np.int32(v)
    ts = env.step(flat)
    acts.append(ids)
    rews.append([float(ts.observation[f"{p + 1}.REWARD"]) for p in range(a.players)])
    snap(ts)
  extra = {"position": np.array(pos), "orientation": np.array(ori)} if pos else {}
  np.savez_compressed(a.out, actions=np.array(acts), rewards=np.array(rews),
                      world_rgb=np.array(world), rgb=np.array(rgb), seed=a.seed, **extra)
  print("wrote", a.out)


if __name__ == "__main__":
  main()
