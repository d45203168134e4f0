"""GPU parity for coop_mining (a sixth Lua level: lua/levels/coop_mining/components.lua):
the HIP engine through the C ABI against the CPU oracle, bit-exact on the grid, the ores'
hidden Lua-side variables (miners, countdowns: packed into the state dump), the avatars, f64
rewards, READY_TO_SHOOT, events and every RGB byte of both views — on the stock pack and on
one whose ore grows 100 x faster (random play meets gold co-mining, window timeouts and two
beams on one ore only there)."""
import numpy as np
import pytest

import util
from test_gpu_parity import _compare_rgb, _compare_scalars, _compare_state, _engine, _run
from test_oracle_coop_cpu import EXTRACTION, MINING, PAIR, rich

pytestmark = pytest.mark.gpu

MINE_HEAVY = [1, 3, 1, 1, 1, 2, 2, 5]     # weights over the ACTION_SET: a third of the actions mine


@pytest.mark.parametrize("fused", ["agents", "world", "both", None])
def test_short_rollouts_in_every_launch_form(coop_mining_pack, fused):
  _run(rich(coop_mining_pack), n=8, steps=80, seed=5, weights=MINE_HEAVY, rgb_every=8, fused=fused)


def test_unfused_launches_give_the_same_results(coop_mining_pack):
  _run(rich(coop_mining_pack), n=6, steps=40, seed=6, weights=MINE_HEAVY, rgb_every=5, fused="both",
       unfused=True)


def test_1000_fixed_seed_steps(coop_mining_pack):
  """64 worlds x 1000 steps on the stock pack (ore is rare: 2e-4 / 8e-5 per site and frame),
  6 players, state every 10 steps, pixels every 100."""
  _run(coop_mining_pack, n=64, steps=1000, seed=11, weights=MINE_HEAVY, rgb_every=100, state_every=10)


def test_1000_steps_with_plentiful_ore_and_other_player_counts(coop_mining_pack):
  pk = rich(coop_mining_pack)
  _run(pk, n=48, steps=1000, seed=12, weights=MINE_HEAVY, rgb_every=100, state_every=5)
  _run(pk, n=16, steps=300, seed=13, weights=MINE_HEAVY, rgb_every=50, state_every=5, num_players=8)
  _run(pk, n=16, steps=300, seed=14, weights=MINE_HEAVY, rgb_every=50, state_every=5, num_players=2)


def test_events_and_the_rules_they_report(coop_mining_pack):
  """Every world, every step: the event rows are the oracle's (as a multiset), the three
  kinds all occur, and what they report is what was paid."""
  import torch
  from meltingpot_amd import engine as E
  pk = rich(coop_mining_pack, iron=0.03, gold=0.05)
  n, steps = 40, 400
  eng = _engine(pk, n)
  eng.bind(E.OBS_RGB)
  oracles = util.make_oracles(pk, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(3)
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions, weights=MINE_HEAVY)
  seen = {MINING: 0, EXTRACTION: 0, PAIR: 0}
  for s in range(steps):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    ev = eng.observe(E.OBS_EVENTS).cpu().numpy()
    rew = eng.observe(E.OBS_REWARD).cpu().numpy()
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
      got = sorted(tuple(int(v) for v in r[:3]) for r in ev[w, 1:1 + int(ev[w, 0, 0])])
      assert got == sorted(o.events()), (s, w)
      assert np.array_equal(rew[w], o.rewards()), (s, w)
      for t, _, _ in got:
        seen[t] += 1
  assert seen[MINING] > 500 and seen[EXTRACTION] > 300 and seen[PAIR] >= 10, seen
  names = {name for w in range(4) for name, _ in eng.events(w)}
  assert names <= {"mining", "extraction", "extraction_pair"}
  _compare_state(eng, oracles, "end")
  _compare_rgb(eng, oracles, "end")
  eng.close()


@pytest.mark.parametrize("n,groups,auto_reset", [(100, 4, False), (90, 2, True)])
def test_fused_ring_recycles_buffers(coop_mining_pack, n, groups, auto_reset):
  """The fused launch with many batches per workgroup (test_gpu_parity.py's case for the other
  levels), worlds restarting inside the ring in the second case: a reset's frame 0 runs the
  regrow updaters too."""
  import torch
  from meltingpot_amd import engine as E
  pk = rich(coop_mining_pack, iron=0.05, gold=0.05)
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
  # (at these rates some worlds start with ore: the grid:update of api:start grows it)
  assert sum(int(o.dump()[2][3]) for o in oracles) > 0
  rng = np.random.default_rng(n)
  acts = util.random_actions(rng, 30, n, eng.P, eng.num_actions, weights=MINE_HEAVY)
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


def test_substrate_api(coop_mining_pack):
  """`substrate.build("coop_mining", roles=..., num_worlds=N)`: specs, both roles, the batched
  timestep against the oracle."""
  import torch
  from meltingpot_amd import substrate
  from oracle import oracle as oracle_lib
  cfg = substrate.get_config("coop_mining")
  assert cfg.valid_roles == frozenset({"default", "target"}) and len(cfg.default_player_roles) == 6
  env = substrate.build("coop_mining", roles=("default", "target") * 3, num_worlds=5, env_seed=300)
  spec = env.observation_spec()[0]
  assert spec["WORLD.RGB"].shape == (216, 216, 3) and spec["RGB"].shape == (88, 88, 3)
  assert env.action_spec()[0].num_values == 8
  refs = [oracle_lib.Oracle(coop_mining_pack, 300 + w, 6) for w in range(5)]
  ts = env.reset()
  for o in refs:
    o.reset()
  rng = np.random.default_rng(1)
  for _ in range(25):
    a = rng.integers(0, 8, size=(5, 6)).astype(np.int32)
    ts = env.step(torch.from_numpy(a).to(env.engine.device))
    for w, o in enumerate(refs):
      o.step(a[w])
      assert np.array_equal(ts.observation["WORLD.RGB"][w].cpu().numpy(), o.render_world())
      assert np.array_equal(ts.observation["READY_TO_SHOOT"][w].cpu().numpy(), o.ready_to_shoot())
      assert np.array_equal(ts.reward[w].cpu().numpy(), o.rewards())
  env.close()
