"""The shuffled visiting orders a step leaves in the world's record for the NEXT step
(WorldTail::next_orders, csrc/step_common.h: step_orders / finish) are the orders that step
would have drawn itself: an engine that leaves them (the product) and one that does not
(MpDevOptions.no_next_orders) stay bit-identical through masked resets with new seeds, episode
ends with auto-reset, and snapshot / restore — in the fused launch and in the stand-alone step
kernels.  (That the product's results are the ORACLE's is every other GPU test.)"""
import numpy as np
import pytest

import util
from test_gpu_parity import _engine

pytestmark = pytest.mark.gpu

LEVELS = ["clean_up", "commons_harvest__open", "territory__rooms", "coins",
          "prisoners_dilemma_in_the_matrix__arena", "coop_mining", "gift_refinements",
          "collaborative_cooking__crowded", "externality_mushrooms__dense"]


def _same(a, b, tag):
  from meltingpot_amd import engine as E
  for x, y, what in zip(a.dump(), b.dump(), ("grid", "avatars", "globals")):
    assert np.array_equal(x, y), (tag, what)
  for kind in (E.OBS_REWARD, E.OBS_READY_TO_SHOOT, E.OBS_STEP_TYPE, E.OBS_EVENTS):
    assert np.array_equal(a.observe_host(kind), b.observe_host(kind)), (tag, kind)


@pytest.mark.parametrize("fused", ["world", None])
@pytest.mark.parametrize("name", LEVELS)
def test_left_orders_are_the_drawn_orders(name, fused):
  import torch
  from meltingpot_amd import engine as E
  # short episodes: some worlds end and restart (auto-reset) inside the run
  pack = util.patch_pack(E.load_pack(name), MAXFRAMES=23)
  n, steps = 70, 64
  a = _engine(pack, n, auto_reset=True, dev={"static_pct": 100})
  b = _engine(pack, n, auto_reset=True, dev={"static_pct": 100, "no_next_orders": 1})
  views = [e.bind(E.OBS_WORLD_RGB) if fused else None for e in (a, b)]
  a.reset(); b.reset()
  rng = np.random.default_rng(7)
  acts = util.random_actions(rng, steps, n, a.P, a.num_actions)
  snaps, at_snap = None, None
  for s in range(steps):
    if s == 11:     # every third world restarts under another seed
      mask = (np.arange(n) % 3 == 0).astype(np.uint8)
      seeds = np.arange(n, dtype=np.uint64) * 7919 + 5
      a.reset(seeds, mask); b.reset(seeds, mask)
      _same(a, b, "masked reset")
    if s == 30:
      snaps = (a.snapshot(), b.snapshot())
    t = torch.from_numpy(acts[s]).to(a.device)
    a.step(t); b.step(t)
    _same(a, b, s)
    if fused:
      assert torch.equal(views[0], views[1]), s
    if s == 40:
      at_snap = a.dump()
  # back to step 30, the same actions again: the same worlds at step 40 — from either
  # engine's snapshot in either engine (a record with orders in an engine that leaves none,
  # and the other way round)
  a.restore(snaps[1]); b.restore(snaps[0])
  for s in range(30, 41):
    t = torch.from_numpy(acts[s]).to(a.device)
    a.step(t); b.step(t)
  for e in (a, b):
    for x, y in zip(e.dump(), at_snap):
      assert np.array_equal(x, y)
  a.close(); b.close()
