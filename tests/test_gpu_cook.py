"""GPU parity for collaborative_cooking (an eighth Lua level, seven substrates:
lua/levels/collaborative_cooking/components.lua): the HIP engine through the C ABI against the
CPU oracle, bit-exact on the grid (the inventories' states with the facing of an avatar's, the
pots, the loading bars, every avatar's interact sprite), the pots' cooking times, f64 rewards,
events and every RGB byte of both views — on the seven stock packs and on kitchens with
something on every counter and a short cooking time (random play cooks and delivers only there)."""
import numpy as np
import pytest

import util
from test_gpu_parity import _compare_rgb, _compare_scalars, _compare_state, _engine, _run
from test_oracle_cook_cpu import ACCEPTED, COLLECTED, DROPPED, INTERACT_HEAVY, LAYOUTS, stocked

pytestmark = pytest.mark.gpu


def _pack(layout):
  from meltingpot_amd import engine
  return engine.load_pack(f"collaborative_cooking__{layout}")


@pytest.mark.parametrize("fused", ["agents", "world", "both", None])
def test_short_rollouts_in_every_launch_form(fused):
  _run(stocked(_pack("cramped")), n=8, steps=120, seed=5, weights=INTERACT_HEAVY, rgb_every=8, fused=fused)


def test_unfused_launches_give_the_same_results():
  _run(stocked(_pack("circuit")), n=6, steps=60, seed=6, weights=INTERACT_HEAVY, rgb_every=5, fused="both",
       unfused=True)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_1000_fixed_seed_steps_of_every_layout(layout):
  """48 worlds x 1000 steps — one whole episode and its end — on the stock pack, the config's
  player count, state every 10 steps, pixels every 100; then 400 steps on the stocked kitchen."""
  _run(_pack(layout), n=48, steps=1000, seed=11, weights=INTERACT_HEAVY, rgb_every=100, state_every=10)
  _run(stocked(_pack(layout)), n=32, steps=400, seed=12, weights=INTERACT_HEAVY, rgb_every=50, state_every=5)


def test_fewer_players_than_the_pack_holds():
  _run(stocked(_pack("crowded")), n=16, steps=300, seed=13, weights=INTERACT_HEAVY, rgb_every=50, state_every=5,
       num_players=4)
  _run(stocked(_pack("figure_eight")), n=16, steps=300, seed=14, weights=INTERACT_HEAVY, rgb_every=50,
       state_every=5, num_players=1)


@pytest.mark.parametrize("layout", ["cramped", "crowded"])
def test_events_and_rewards_every_step(layout):
  """Every world, every step: the event rows are the oracle's (as a multiset), the rewards too;
  ingredients are dropped, soups collected and delivered."""
  import torch
  from meltingpot_amd import engine as E
  pk = stocked(_pack(layout))
  n, steps = 64, 600
  eng = _engine(pk, n)
  eng.bind(E.OBS_RGB)
  oracles = util.make_oracles(pk, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(3)
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions, weights=INTERACT_HEAVY)
  seen = {ACCEPTED: 0, DROPPED: 0, COLLECTED: 0}
  paid = 0
  for s in range(steps):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    ev = eng.observe(E.OBS_EVENTS).cpu().numpy()
    rew = eng.observe(E.OBS_REWARD).cpu().numpy()
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
      got = sorted(tuple(int(v) for v in r[:3]) for r in ev[w, 1:1 + int(ev[w, 0, 0])])
      assert got == sorted(o.events()), (s, w)
      assert np.array_equal(rew[w], o.rewards()), (s, w)
      paid += int(rew[w].sum() > 0)
      for t, _, _ in got:
        seen[t] += 1
  assert seen[DROPPED] >= 50 and seen[COLLECTED] >= 1 and seen[ACCEPTED] >= 10 and paid >= 10, (seen, paid)
  names = {name for w in range(8) for name, _ in eng.events(w)}
  assert names <= {"receiver_accepted_item", "item_dropped_into_pot", "cooked_food_collected_from_pot"}
  _compare_state(eng, oracles, "end")
  _compare_rgb(eng, oracles, "end")
  eng.close()


@pytest.mark.parametrize("n,groups,auto_reset", [(100, 4, False), (90, 2, True)])
def test_fused_ring_recycles_buffers(n, groups, auto_reset):
  """The fused launch with many batches per workgroup, worlds restarting inside the ring in the
  second case (counters re-stocked, pots emptied, inventories back on their avatars)."""
  import torch
  from meltingpot_amd import engine as E
  pk = stocked(_pack("ring"))
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
  acts = util.random_actions(rng, 30, n, eng.P, eng.num_actions, weights=INTERACT_HEAVY)
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


def test_substrate_api():
  """`substrate.build("collaborative_cooking__cramped", ...)`: specs, the batched timestep against
  the oracle, the decoded events."""
  import torch
  from meltingpot_amd import substrate
  from oracle import oracle as oracle_lib
  name = "collaborative_cooking__cramped"
  cfg = substrate.get_config(name)
  assert cfg.valid_roles == frozenset({"default"}) and len(cfg.default_player_roles) == 2
  env = substrate.build(name, roles=("default",) * 2, num_worlds=5, env_seed=300)
  spec = env.observation_spec()[0]
  assert spec["WORLD.RGB"].shape == (40, 72, 3) and spec["RGB"].shape == (40, 40, 3)
  assert env.action_spec()[0].num_values == 8
  refs = [oracle_lib.Oracle(_pack("cramped"), 300 + w, 2) for w in range(5)]
  ts = env.reset()
  for o in refs:
    o.reset()
  rng = np.random.default_rng(1)
  for _ in range(40):
    a = rng.integers(0, 8, size=(5, 2)).astype(np.int32)
    ts = env.step(torch.from_numpy(a).to(env.engine.device))
    for w, o in enumerate(refs):
      o.step(a[w])
      assert np.array_equal(ts.observation["WORLD.RGB"][w].cpu().numpy(), o.render_world())
      for p in range(2):
        assert np.array_equal(ts.observation["RGB"][w, p].cpu().numpy(), o.render_agent(p))
      assert np.array_equal(ts.reward[w].cpu().numpy(), o.rewards())
  env.close()
