"""GPU parity for gift_refinements (a seventh Lua level:
lua/levels/gift_refinements/components.lua): the HIP engine through the C ABI against the CPU
oracle, bit-exact on the grid, the avatars' Lua-side variables (inventories, both timers:
packed into the state dump), f64 rewards, READY_TO_SHOOT, the INVENTORY observation, events
and every RGB byte of both views — on the stock pack and on one whose tokens grow 100 x faster
(random play meets picks, refinements, gifts at capacity and two gifts on one avatar only there)."""
import numpy as np
import pytest

import util
from test_gpu_parity import _compare_rgb, _compare_scalars, _compare_state, _engine, _run
from test_oracle_gift_cpu import GIFT, GIFT_HEAVY, decode_gift, rich

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gift_pack() -> bytes:
  from meltingpot_amd import engine
  return engine.load_pack("gift_refinements")


@pytest.mark.parametrize("fused", ["agents", "world", "both", None])
def test_short_rollouts_in_every_launch_form(gift_pack, fused):
  _run(rich(gift_pack), n=8, steps=80, seed=5, weights=GIFT_HEAVY, rgb_every=8, fused=fused)


def test_unfused_launches_give_the_same_results(gift_pack):
  _run(rich(gift_pack), n=6, steps=40, seed=6, weights=GIFT_HEAVY, rgb_every=5, fused="both",
       unfused=True)


def test_1000_fixed_seed_steps(gift_pack):
  """64 worlds x 1000 steps on the stock pack (tokens are rare: 2e-4 per site and frame),
  6 players, state every 10 steps, pixels every 100."""
  _run(gift_pack, n=64, steps=1000, seed=11, weights=GIFT_HEAVY, rgb_every=100, state_every=10)


def test_1000_steps_with_plentiful_tokens_and_other_player_counts(gift_pack):
  pk = rich(gift_pack)
  _run(pk, n=48, steps=1000, seed=12, weights=GIFT_HEAVY, rgb_every=100, state_every=5)
  _run(pk, n=16, steps=300, seed=13, weights=GIFT_HEAVY, rgb_every=50, state_every=5, num_players=8)
  _run(pk, n=16, steps=300, seed=14, weights=GIFT_HEAVY, rgb_every=50, state_every=5, num_players=2)


def test_events_inventories_and_rewards_every_step(gift_pack):
  """Every world, every step: the event rows are the oracle's (as a multiset), the INVENTORY
  observation and the rewards are the oracle's; gifts of every source type occur, and so do
  gifts that leave the recipient at the capacity of 15."""
  import torch
  from meltingpot_amd import engine as E
  pk = rich(gift_pack, rate=0.05)
  n, steps = 40, 500
  eng = _engine(pk, n)
  eng.bind(E.OBS_RGB)
  assert eng.info.num_resources == 3
  oracles = util.make_oracles(pk, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(3)
  # (few consumptions: inventories fill up)
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions, weights=[1, 4, 1, 1, 1, 2, 2, 6, 0.2])
  by_src = [0, 0, 0]
  at_capacity = 0
  for s in range(steps):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    ev = eng.observe(E.OBS_EVENTS).cpu().numpy()
    rew = eng.observe(E.OBS_REWARD).cpu().numpy()
    inv = eng.observe(E.OBS_INVENTORY).cpu().numpy()
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
      got = sorted(tuple(int(v) for v in r[:3]) for r in ev[w, 1:1 + int(ev[w, 0, 0])])
      assert got == sorted(o.events()), (s, w)
      assert np.array_equal(rew[w], o.rewards()), (s, w)
      assert np.array_equal(inv[w], o.inventories()[0]), (s, w)
      for t, a, b in got:
        if t == GIFT:
          _, src, _, cnt = decode_gift(a, b)
          by_src[src] += 1
          at_capacity += cnt == 15
  assert min(by_src) >= 5 and at_capacity >= 1, (by_src, at_capacity)
  names = {name for w in range(4) for name, _ in eng.events(w)}
  assert names <= {"gift"}
  _compare_state(eng, oracles, "end")
  _compare_rgb(eng, oracles, "end")
  eng.close()


@pytest.mark.parametrize("n,groups,auto_reset", [(100, 4, False), (90, 2, True)])
def test_fused_ring_recycles_buffers(gift_pack, n, groups, auto_reset):
  """The fused launch with many batches per workgroup (test_gpu_parity.py's case for the other
  levels), worlds restarting inside the ring in the second case (inventories and timers start
  empty again)."""
  import torch
  from meltingpot_amd import engine as E
  pk = rich(gift_pack, rate=0.05)
  if auto_reset:
    pk = util.patch_pack(pk, MAXFRAMES=9)
  eng = _engine(pk, n, auto_reset=auto_reset, unfused=False, dev={"max_groups": groups})
  eng.bind(E.OBS_RGB)
  assert eng.fused
  oracles = util.make_oracles(pk, n)
  eng.reset()
  for o in oracles:
    o.reset()
  _compare_state(eng, oracles, "reset")
  _compare_rgb(eng, oracles, "reset")
  rng = np.random.default_rng(n)
  acts = util.random_actions(rng, 30, n, eng.P, eng.num_actions, weights=GIFT_HEAVY)
  restarts = 0
  for s in range(30):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      if o.done and auto_reset:
        o.reset(); restarts += 1
      else:
        o.step(acts[s, w])
    if s % 4 == 3 or s == 29:
      _compare_state(eng, oracles, f"step {s + 1}")
      _compare_scalars(eng, oracles, f"step {s + 1}")
      _compare_rgb(eng, oracles, f"step {s + 1}")
  assert restarts >= (2 * n if auto_reset else 0)
  assert not eng.fault_words()[:6].any()
  eng.close()


def test_substrate_api(gift_pack):
  """`substrate.build("gift_refinements", roles=..., num_worlds=N)`: specs, both roles, the
  batched timestep (INVENTORY included) against the oracle, and the decoded `gift` event."""
  import torch
  from meltingpot_amd import substrate
  from oracle import oracle as oracle_lib
  cfg = substrate.get_config("gift_refinements")
  assert cfg.valid_roles == frozenset({"default", "target"}) and len(cfg.default_player_roles) == 6
  env = substrate.build("gift_refinements", roles=("default", "target") * 3, num_worlds=5, env_seed=300)
  spec = env.observation_spec()[0]
  assert spec["WORLD.RGB"].shape == (216, 216, 3) and spec["RGB"].shape == (88, 88, 3)
  assert spec["INVENTORY"].shape == (3,)
  assert env.action_spec()[0].num_values == 9
  refs = [oracle_lib.Oracle(gift_pack, 300 + w, 6) for w in range(5)]
  ts = env.reset()
  for o in refs:
    o.reset()
  rng = np.random.default_rng(1)
  for _ in range(25):
    a = rng.integers(0, 9, size=(5, 6)).astype(np.int32)
    ts = env.step(torch.from_numpy(a).to(env.engine.device))
    for w, o in enumerate(refs):
      o.step(a[w])
      assert np.array_equal(ts.observation["WORLD.RGB"][w].cpu().numpy(), o.render_world())
      assert np.array_equal(ts.observation["READY_TO_SHOOT"][w].cpu().numpy(), o.ready_to_shoot())
      assert np.array_equal(ts.observation["INVENTORY"][w].cpu().numpy(), o.inventories()[0])
      assert np.array_equal(ts.reward[w].cpu().numpy(), o.rewards())
  env.close()
