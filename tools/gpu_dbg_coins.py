"""dev: coins WORLD.RGB draw-only with few workgroups vs the oracle (where, how often)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import util
from meltingpot_amd import engine as E
name, n, groups = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
pack = E.load_pack(name)
for bind in (True, False):
  eng = E.Engine(pack, n, unfused=False, dev={"max_groups": groups, "verbose": 1} if groups else None)
  if bind:
    eng.bind(E.OBS_RGB)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles: o.reset()
  want = np.stack([o.render_world() for o in oracles])
  for rep in range(4):
    got = eng.observe(E.OBS_WORLD_RGB).cpu().numpy()
    bad = np.argwhere((got != want).any(axis=-1))
    print(f"{name} n={n} groups={groups} bound={bind} rep={rep}: {len(bad)} bad pixels",
          "worlds", sorted(set(bad[:, 0].tolist()))[:12],
          "first", bad[:3].tolist())
  eng.close()
