"""Every committed pack through the fused launch with BOTH views bound and many batches per
workgroup (24 worlds on each of 8 workgroups: the ring of LDS buffers is recycled again and
again), 250 steps of uniformly random actions, against the oracle: state and scalars every 25
steps, every RGB byte of both views every 50.  A world whose episode has ended restarts on the
next step (auto-reset), as its oracle does.  One test per substrate the package registers."""
import zlib

import numpy as np
import pytest

import util
from test_gpu_parity import _compare_rgb, _compare_scalars, _compare_state, _engine

pytestmark = pytest.mark.gpu


def _names():
  from meltingpot_amd import substrate
  return sorted(substrate.SUBSTRATES)


@pytest.mark.parametrize("form", ["fused", "stand_alone"])
@pytest.mark.parametrize("name", _names())
def test_soak(name, form):
  """`stand_alone`: no view bound — the stand-alone step kernels, the views drawn by
  mp_observe's draw-only launches — 128 worlds, 150 steps."""
  import torch
  from meltingpot_amd import engine as E
  pack = E.load_pack(name)
  n, steps = (192, 250) if form == "fused" else (128, 150)
  if form == "fused":
    eng = _engine(pack, n, auto_reset=True, unfused=False, dev={"max_groups": 8})
    eng.bind(E.OBS_RGB); eng.bind(E.OBS_WORLD_RGB)
    assert eng.fused
  else:
    eng = _engine(pack, n, auto_reset=True)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  _compare_state(eng, oracles, "reset")
  _compare_rgb(eng, oracles, "reset")
  rng = np.random.default_rng(zlib.crc32((name + form).encode()))
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions)
  dacts = torch.from_numpy(acts).to(eng.device)
  for s in range(steps):
    eng.step(dacts[s])
    for w, o in enumerate(oracles):
      if o.done:
        o.reset()
      else:
        o.step(acts[s, w])
    if (s + 1) % 25 == 0:
      _compare_state(eng, oracles, f"step {s + 1}")
      _compare_scalars(eng, oracles, f"step {s + 1}")
    if (s + 1) % 50 == 0:
      _compare_rgb(eng, oracles, f"step {s + 1}")
  assert not eng.fault_words()[:6].any()
  eng.close()


@pytest.mark.parametrize("draw", [0, 1, 2, 3])
@pytest.mark.parametrize("name", _names())
def test_fuzz(name, draw):
  """Four configurations per substrate, drawn from the substrate's name: which views are bound, how
  many worlds (37 / 130 / 300), how many players (where the level allows fewer than the pack
  holds), a random mix of actions, and — a third of the way in — a masked reset of random worlds
  under new seeds; a world whose episode ends restarts on the next step, as its oracle does."""
  import torch
  from meltingpot_amd import engine as E
  from oracle import oracle as oracle_lib
  pack = E.load_pack(name)
  rng = np.random.default_rng(zlib.crc32((f"fuzz{draw}" + name).encode()))
  form = ("agents", "world", "both", None)[rng.integers(4)]
  n = (37, 130, 300)[rng.integers(3)]
  kw = {}
  fixed_players = "in_the_matrix" in name or name == "coins"
  if not fixed_players:
    from meltingpot_amd import pack as pack_lib, lower
    p_pack = int(pack_lib.loads(pack)["hdr"][lower.HDR_P])
    kw["num_players"] = int(rng.integers(1, p_pack + 1))
  eng = _engine(pack, n, **kw)
  if form in ("agents", "both"):
    eng.bind(E.OBS_RGB)
  if form in ("world", "both"):
    eng.bind(E.OBS_WORLD_RGB)
  P = eng.P
  oracles = [oracle_lib.Oracle(pack, util.world_seed(w), kw.get("num_players", 0)) for w in range(n)]
  eng.reset()
  for o in oracles:
    o.reset()
  steps = 180
  weights = rng.dirichlet(np.ones(eng.num_actions) * 0.7) + 0.02
  acts = util.random_actions(rng, steps, n, P, eng.num_actions, weights=weights)
  dacts = torch.from_numpy(acts).to(eng.device)
  for s in range(steps):
    if s == steps // 3:
      mask = (rng.random(n) < 0.3).astype(np.uint8)
      seeds = rng.integers(1, 1 << 62, size=n, dtype=np.uint64)
      eng.reset(seeds, mask)
      for w in np.flatnonzero(mask):
        oracles[w].close()
        oracles[w] = oracle_lib.Oracle(pack, int(seeds[w]), kw.get("num_players", 0))
        oracles[w].reset()
      _compare_state(eng, oracles, f"masked reset ({form}, n={n}, P={P})")
      _compare_rgb(eng, oracles, f"masked reset ({form}, n={n}, P={P})")
    eng.step(dacts[s])
    for w, o in enumerate(oracles):
      if o.done:
        o.reset()
      else:
        o.step(acts[s, w])
    if (s + 1) % 30 == 0:
      _compare_state(eng, oracles, f"step {s + 1} ({form}, n={n}, P={P})")
      if (s + 1) % 60 == 0:
        _compare_rgb(eng, oracles, f"step {s + 1} ({form}, n={n}, P={P})")
  assert not eng.fault_words()[:6].any()
  eng.close()


@pytest.mark.parametrize("name", _names())
def test_shards_and_snapshots(name):
  """Per substrate, without the oracle: 40 worlds in one engine are the 20 + 20 worlds of two
  engines with `world_offset` (results do not depend on how the worlds are sharded); a snapshot
  taken mid-run and restored into a FRESH engine continues to the same state, rewards and
  pixels (the record carries everything a level keeps between steps: hidden planes, timers,
  inventories, the next step's orders)."""
  import torch
  from meltingpot_amd import engine as E
  pack = E.load_pack(name)
  n = 40
  whole = _engine(pack, n)
  lo, hi = _engine(pack, n // 2), _engine(pack, n // 2, world_offset=n // 2)
  for e in (whole, lo, hi):
    e.bind(E.OBS_WORLD_RGB)
    e.reset()
  rng = np.random.default_rng(zlib.crc32(("shards" + name).encode()))
  acts = torch.from_numpy(util.random_actions(rng, 40, n, whole.P, whole.num_actions)).to(whole.device)
  snap = None
  for s in range(30):
    if s == 20:
      snap = whole.snapshot()
    whole.step(acts[s]); lo.step(acts[s, :n // 2].contiguous()); hi.step(acts[s, n // 2:].contiguous())
  for a, b, c in zip(whole.dump(), lo.dump(), hi.dump()):
    assert np.array_equal(a, np.concatenate([b, c]))
  assert np.array_equal(whole.observe_host(E.OBS_REWARD),
                        np.concatenate([lo.observe_host(E.OBS_REWARD), hi.observe_host(E.OBS_REWARD)]))
  want = whole.dump(), whole.observe_host(E.OBS_REWARD), whole.observe_host(E.OBS_RGB), whole.observe_host(E.OBS_WORLD_RGB)
  fresh = _engine(pack, n)
  fresh.reset()
  fresh.restore(snap)
  for s in range(20, 30):
    fresh.step(acts[s])
  for a, b in zip(fresh.dump(), want[0]):
    assert np.array_equal(a, b)
  assert np.array_equal(fresh.observe_host(E.OBS_REWARD), want[1])
  assert np.array_equal(fresh.observe_host(E.OBS_RGB), want[2])
  assert np.array_equal(fresh.observe_host(E.OBS_WORLD_RGB), want[3])
  for e in (whole, lo, hi, fresh):
    e.close()
