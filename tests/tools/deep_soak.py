"""dev helper (GPU box): every committed pack at scale and over a long horizon against the
oracle — N worlds (default 2048) stepped `steps` times (default 900: only coins and the one-shot
matrix games end episodes that early — those worlds then stay as they are, no auto-reset, and a
step asked of them reports nothing: round 6 found the oracle repeating the last step's reward
there, world 835 of coins) by the fused launch with a view bound, actions a pure function of
(global world, step, player) (tests/util.py:hashed_actions), and SAMPLED worlds — six blocks
of sixteen, anywhere in the batch — replayed by the oracle from the same function: state,
hidden rule variables, rewards and events after the last step, the bound view after every
third of the run.  What the 8-world soak (tests/tools/soak.py) cannot meet — the one world in
a thousand in which a rare rule fires — this can (round 6: 21 of 16384 externality_mushrooms
worlds put a marking beside its avatar).

  python tests/tools/deep_soak.py [worlds] [steps] [name filters...]"""
import glob
import os
import sys
import time

_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(_TESTS))
sys.path.insert(0, _TESTS)
import numpy as np
import torch
import util
from meltingpot_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 900
SUBS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(
    os.path.dirname(_TESTS), "meltingpot_amd", "assets", "*.mpk")))
if len(sys.argv) > 3:
  SUBS = [x for x in SUBS if any(k in x for k in sys.argv[3:])]
looks = tuple(sorted({steps // 3, 2 * steps // 3, steps}))
failures = []
for isub, sub in enumerate(SUBS):
  t0 = time.time()
  pack = E.load_pack(sub)
  eng = E.Engine(pack, n, device=0, auto_reset=False, unfused=False, placements=1)
  world_view = isub % 2 == 0
  kind = E.OBS_WORLD_RGB if world_view else E.OBS_RGB
  bound = eng.bind(kind)
  rng = np.random.default_rng(1000 + isub)
  blocks = sorted(int(b) for b in rng.choice(n // 16, size=min(6, n // 16), replace=False) * 16)
  worlds = np.arange(n)
  eng.reset()
  seen = {}
  for s in range(steps):
    eng.step(torch.from_numpy(util.hashed_actions(worlds, s, eng.P, num_actions=eng.num_actions)).to(eng.device))
    if s + 1 in looks:
      seen[s + 1] = {b: bound[b:b + 16].cpu().numpy() for b in blocks}
  grid, avat, glob_ = eng.dump()
  rew = eng.observe(E.OBS_REWARD).cpu().numpy()
  ev = eng.observe(E.OBS_EVENTS).cpu().numpy()
  ended = int((glob_[:, 1] != 0).sum())
  bad = 0
  for b in blocks:
    acts = np.stack([util.hashed_actions(range(b, b + 16), s, eng.P, num_actions=eng.num_actions)
                     for s in range(steps)])
    for w, og, oa, ogl, orew, oev, views in util.replay_parallel(
        pack, acts, looks=looks, sample=range(b, b + 16), world_view=world_view, offset=b, workers=8):
      ok = (np.array_equal(grid[w], og) and np.array_equal(avat[w], oa) and np.array_equal(glob_[w], ogl)
            and np.array_equal(rew[w], orew))
      got = sorted(tuple(int(v) for v in r[:3]) for r in ev[w, 1:1 + int(ev[w, 0, 0])])
      ok = ok and got == sorted(oev)
      for look, view in views.items():
        ok = ok and np.array_equal(seen[look][b][w - b], view)
      if not ok:
        bad += 1
        failures.append((sub, w))
  print(f"{sub}: {n} worlds x {steps} steps, {'WORLD.RGB' if world_view else 'RGB'} bound, worlds "
        f"{[f'{b}-{b + 15}' for b in blocks]} replayed: {'ok' if not bad else f'{bad} WORLDS DIFFER'} "
        f"({ended} worlds ended, counters {eng.counters()}, {time.time() - t0:.1f} s)", flush=True)
  eng.close()
print("deep soak:", "all packs ok" if not failures else f"DIFFERENCES in {failures}")
sys.exit(1 if failures else 0)
