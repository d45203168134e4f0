"""dev helper: long auto-reset rollout of every pack on the GPU against the oracle
(natural episode ends of StochasticIntervalEpisodeEnding included).
usage: python tests/tools/soak.py [steps] [worlds] [name filters...]"""
import os, sys
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(_TESTS))
sys.path.insert(0, _TESTS)
import numpy as np
import torch
import util
from meltingpot_amd import engine as E

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
import glob
SUBS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(
    os.path.dirname(_TESTS), "meltingpot_amd", "assets", "*.mpk")))   # every committed pack
if len(sys.argv) > 3:
  SUBS = [x for x in SUBS if any(k in x for k in sys.argv[3:])]
for isub, sub in enumerate(SUBS):
  pack = E.load_pack(sub)
  eng = E.Engine(pack, n, device=0, auto_reset=True, unfused=False)
  # the fused launch steps the worlds and draws the bound view; the other view is
  # drawn from the stepped records by mp_observe
  bound_kind = E.OBS_WORLD_RGB if isub % 2 == 0 else E.OBS_RGB
  bound = eng.bind(bound_kind)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(1)
  episodes = 0
  for s in range(steps):
    acts = rng.integers(0, eng.num_actions, size=(n, eng.P), dtype=np.int32)
    eng.step(torch.from_numpy(acts).to(eng.device))
    for w, o in enumerate(oracles):
      if o.done:
        o.reset(); episodes += 1
      else:
        o.step(acts[w])
    if s % 25 == 0 or s == steps - 1:
      grid, avat, glob = eng.dump()
      rew = eng.observe(E.OBS_REWARD).cpu().numpy()
      for w, o in enumerate(oracles):
        og, oa, ogl = o.dump()
        assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa), (sub, s, w)
        assert np.array_equal(glob[w], ogl), (sub, s, w, glob[w], ogl)
        if not o.done and o._L.orc_step_count(o._h) > 0:
          assert np.array_equal(rew[w], o.rewards()), (sub, s, w)
    if s % 500 == 0 or s == steps - 1:
      rgb = (bound if bound_kind == E.OBS_RGB else eng.observe(E.OBS_RGB)).cpu().numpy()
      wrgb = (bound if bound_kind == E.OBS_WORLD_RGB else eng.observe(E.OBS_WORLD_RGB)).cpu().numpy()
      for w, o in enumerate(oracles):
        assert np.array_equal(wrgb[w], o.render_world()), (sub, s, w)
        for p in range(o.P):
          assert np.array_equal(rgb[w, p], o.render_agent(p)), (sub, s, w, p)
  print(f"{sub}: {steps} steps x {n} worlds ok (fused, "
        f"{'WORLD.RGB' if bound_kind == E.OBS_WORLD_RGB else 'RGB'} bound), {episodes} episode restarts",
        flush=True)
  eng.close()
