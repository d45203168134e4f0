#!/usr/bin/env python3
"""dev helper: find the first step where engine and oracle diverge and print it."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
from meltingpot_amd import engine as E

name, n, steps, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
weights = [float(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else None
pack = E.load_pack(name)
eng = E.Engine(pack, n); oracles = util.make_oracles(pack, n)
eng.reset(); [o.reset() for o in oracles]
rng = np.random.default_rng(seed)
acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions, weights)
names = bytes(oracles[0].tables["state_names"]).decode().split("\0")
for s in range(steps):
  eng.step(torch.from_numpy(acts[s]).to(eng.device))
  for w, o in enumerate(oracles): o.step(acts[s, w])
  grid, avat, glob = eng.dump()
  rew = eng.observe(E.OBS_REWARD).cpu().numpy()
  for w, o in enumerate(oracles):
    og, oa, ogl = o.dump()
    bad = not (np.array_equal(grid[w], og) and np.array_equal(avat[w], oa) and np.array_equal(glob[w], ogl) and np.array_equal(rew[w], o.rewards()))
    if bad:
      print("DIVERGED at step", s + 1, "world", w, "actions", acts[s, w])
      print("glob gpu", glob[w], "\nglob orc", ogl)
      for p in range(eng.P):
        if not np.array_equal(avat[w, p], oa[p]):
          print("avatar", p, "gpu", avat[w, p], hex(avat[w, p, 7]), "\n        orc", oa[p], hex(oa[p, 7]))
      for (l, y, x) in np.argwhere(grid[w] != og)[:12]:
        print("grid layer", l, "xy", (x, y), "gpu", names[grid[w, l, y, x]], "| orc", names[og[l, y, x]])
      print("rew gpu", rew[w], "orc", o.rewards())
      sys.exit(1)
print("no divergence in", steps, "steps")
