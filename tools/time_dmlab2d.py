#!/usr/bin/env python3
"""Baseline B2: time the REAL reference — DMLab2D/Lua — on the
host cores of a machine that has the `dmlab2d` wheel and the reference's
Python dependencies.  Not runnable in the build container (no wheel, no
network); ships so that the number can be produced elsewhere and put next to
bench.py's.

  python tools/time_dmlab2d.py [--substrate clean_up] [--players 7]
                               [--steps 1000] [--procs N]

Loop shape = meltingpot/utils/evaluation/evaluation.py:37-49: reset, then step
with uniformly random actions from ACTION_SET, reading every observation each
step.  One process per core (a Lab2d env is single-threaded).  Prints one JSON
line: agent-steps/s aggregate and per core, core count.
"""
import argparse
import json
import multiprocessing as mp
import os
import time


def _worker(args):
  substrate_name, players, steps, seed = args
  import numpy as np
  from meltingpot import substrate  # the reference package
  config = substrate.get_config(substrate_name)
  roles = ("default",) * players
  env = substrate.build_from_config(config, roles=roles)
  rng = np.random.default_rng(seed)
  n_actions = len(config.action_set)
  env.reset()
  for _ in range(100):  # warm-up
    env.step(rng.integers(0, n_actions, players))
  t0 = time.perf_counter()
  done = 0
  for _ in range(steps):
    ts = env.step(rng.integers(0, n_actions, players))
    done += 1
    if ts.last():
      env.reset()
  dt = time.perf_counter() - t0
  env.close()
  return done, dt


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--substrate", default="clean_up")
  ap.add_argument("--players", type=int, default=7)
  ap.add_argument("--steps", type=int, default=1000)
  ap.add_argument("--procs", type=int, default=os.cpu_count())
  a = ap.parse_args()
  with mp.Pool(a.procs) as pool:
    res = pool.map(_worker, [(a.substrate, a.players, a.steps, 1 + i)
                             for i in range(a.procs)])
  per_core = [a.players * n / dt for n, dt in res]
  print(json.dumps({
      "baseline": "reference (dmlab2d)", "substrate": a.substrate,
      "players": a.players, "cores": a.procs, "steps_per_proc": a.steps,
      "agent_steps_per_s": sum(per_core),
      "agent_steps_per_s_per_core": sum(per_core) / len(per_core)}))


if __name__ == "__main__":
  main()
