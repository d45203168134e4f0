"""round 5: a long zap-heavy rollout of externality_mushrooms__dense (stock pack and a lush one)
on the GPU against the oracle — state compared every 25 steps, rewards and events every step;
counts what the rare marking paths saw (orphans: a marking on the map whose avatar is away;
lost markings: an avatar on the map without one) and MP_CTR_AUX0.
usage: python tools/history/gpu_r05_mushroom_soak.py [steps] [worlds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import util
from meltingpot_amd import engine as E, pack as P
from test_oracle_mushroom_cpu import NAME, ZAP_HEAVY, lush

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
stock = E.load_pack(NAME)
for tag, pk in (("stock", stock), ("lush", lush(stock, seed=21, grow=0.5))):
  t = P.loads(pk)
  mark_layer = int(t["state_layer"][int(t["em_states"][5])])
  eng = E.Engine(pk, n, device=0, auto_reset=True, unfused=False)
  eng.bind(E.OBS_RGB)
  oracles = util.make_oracles(pk, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(7)
  w8 = np.asarray(ZAP_HEAVY, float); w8 /= w8.sum()
  orphans = lost = episodes = 0
  for s in range(steps):
    acts = rng.choice(8, size=(n, eng.P), p=w8).astype(np.int32)
    eng.step(torch.from_numpy(acts).to(eng.device))
    rew = eng.observe(E.OBS_REWARD).cpu().numpy()
    ev = eng.observe(E.OBS_EVENTS).cpu().numpy()
    for w, o in enumerate(oracles):
      if o.done:
        o.reset(); episodes += 1
      else:
        o.step(acts[w])
      assert np.array_equal(rew[w], o.rewards()), (tag, s, w)
      got = sorted(tuple(int(v) for v in r[:3]) for r in ev[w, 1:1 + int(ev[w, 0, 0])])
      assert got == sorted(o.events()), (tag, s, w)
    if s % 25 == 24:
      grid, avat, glob = eng.dump()
      for w, o in enumerate(oracles):
        og, oa, ogl = o.dump()
        assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa) and np.array_equal(glob[w], ogl), (tag, s, w)
        marks = int((og[mark_layer] != 0).sum())
        alive = int(oa[:, 3].sum())
        on_avatar = sum(1 for p in range(o.P) if oa[p, 3] and og[mark_layer, oa[p, 1], oa[p, 0]] != 0)
        orphans += marks - on_avatar
        lost += alive - on_avatar
  c = eng.counters()
  print(f"{tag}: {n} worlds x {steps} steps bit-exact; episodes {episodes}, respawns {c['respawns']}, zaps {c['zaps']}, "
        f"orphaned markings seen {orphans}, avatars without a marking seen {lost} (sampled every 25 steps), aux0 {c['aux0']}")
  eng.close()
