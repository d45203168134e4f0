"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle,
bit-exact on discrete state, rewards/observation scalars and RGB pixels."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _engine(pack, n, **kw):
  import torch
  from meltingpot_amd import engine
  assert torch.cuda.is_available(), "gpu tests need a GPU"
  return engine.Engine(pack, n, **kw)


def _compare_state(eng, oracles, tag):
  grid, avat, glob = eng.dump()
  for w, o in enumerate(oracles):
    og, oa, ogl = o.dump()
    assert np.array_equal(glob[w], ogl), (tag, w, glob[w], ogl)
    if not np.array_equal(avat[w], oa):
      raise AssertionError(f"{tag}: world {w} avatars differ\n{avat[w]}\n{oa}")
    if not np.array_equal(grid[w], og):
      bad = np.argwhere(grid[w] != og)
      raise AssertionError(
          f"{tag}: world {w} grid differs at (layer,y,x)={bad[:8].tolist()} "
          f"gpu={grid[w][tuple(bad[0])]} oracle={og[tuple(bad[0])]}")


def _compare_scalars(eng, oracles, tag):
  from meltingpot_amd import engine as E
  rew = eng.observe(E.OBS_REWARD).cpu().numpy()
  rdy = eng.observe(E.OBS_READY_TO_SHOOT).cpu().numpy()
  aux = eng.observe(E.OBS_AUX0).cpu().numpy()
  col = eng.observe(E.OBS_COLLECTIVE_REWARD).cpu().numpy()
  for w, o in enumerate(oracles):
    assert np.array_equal(rew[w], o.rewards()), (tag, w, rew[w], o.rewards())
    assert np.array_equal(rdy[w], o.ready_to_shoot()), (tag, w)
    assert np.array_equal(aux[w], o.num_others_cleaned()), (tag, w)
    assert col[w] == o.rewards().sum(), (tag, w)


def _compare_rgb(eng, oracles, tag):
  """Both views against the oracle.  A view that is bound to the engine was
  rendered by the step's own launch (the fused k_frame); the other one is drawn
  from the stepped records by mp_observe (the render-only k_frame)."""
  from meltingpot_amd import engine as E
  bound = eng._bound
  rgb = (bound[E.OBS_RGB] if E.OBS_RGB in bound else eng.observe(E.OBS_RGB)).cpu().numpy()
  wrgb = (bound[E.OBS_WORLD_RGB] if E.OBS_WORLD_RGB in bound
          else eng.observe(E.OBS_WORLD_RGB)).cpu().numpy()
  for w, o in enumerate(oracles):
    ow = o.render_world()
    if not np.array_equal(wrgb[w], ow):
      bad = np.argwhere(wrgb[w] != ow)
      raise AssertionError(f"{tag}: WORLD.RGB world {w} differs at {bad[:4].tolist()}")
    for p in range(o.P):
      oa = o.render_agent(p)
      if not np.array_equal(rgb[w, p], oa):
        bad = np.argwhere(rgb[w, p] != oa)
        raise AssertionError(
            f"{tag}: RGB world {w} player {p} differs at {bad[:4].tolist()}: "
            f"gpu={rgb[w, p][tuple(bad[0][:2])]} oracle={oa[tuple(bad[0][:2])]}")


def _run(pack, n, steps, seed, weights=None, rgb_every=10, state_every=1, fused="agents",
         **engine_kw):
  """`fused`: the view bound to the engine, i.e. rendered by the launch that steps
  the worlds ("agents", "world", "both"), or None: stand-alone step kernel."""
  from meltingpot_amd import engine as E
  eng = _engine(pack, n, **engine_kw)
  if fused in ("agents", "both"):
    eng.bind(E.OBS_RGB)
  if fused in ("world", "both"):
    eng.bind(E.OBS_WORLD_RGB)
  oracles = util.make_oracles(pack, n, num_players=engine_kw.get("num_players", 0))
  eng.reset()
  for o in oracles:
    o.reset()
  _compare_state(eng, oracles, "reset")
  _compare_scalars(eng, oracles, "reset")
  _compare_rgb(eng, oracles, "reset")
  rng = np.random.default_rng(seed)
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions, weights)
  import torch
  dacts = torch.from_numpy(acts).to(eng.device)
  for s in range(steps):
    eng.step(dacts[s])
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
    if (s + 1) % state_every == 0 or s == steps - 1:
      _compare_state(eng, oracles, f"step {s + 1}")
      _compare_scalars(eng, oracles, f"step {s + 1}")
    if (s + 1) % rgb_every == 0 or s == steps - 1:
      _compare_rgb(eng, oracles, f"step {s + 1}")
  eng.close()


@pytest.mark.parametrize("fused", ["agents", "world", "both", None])
def test_reset_and_short_rollout(clean_up_pack, fused):
  _run(clean_up_pack, n=8, steps=60, seed=1, rgb_every=5, fused=fused)


def test_unfused_launches_give_the_same_results(clean_up_pack):
  _run(clean_up_pack, n=8, steps=40, seed=3, rgb_every=5, fused="both", unfused=True)


def test_1000_fixed_seed_steps(clean_up_pack):
  """BASELINE.json: bit-exact parity on 1000 fixed-seed steps (64 worlds; WORLD.RGB,
  the benchmarked view, rendered by the fused launch)."""
  _run(clean_up_pack, n=64, steps=1000, seed=1234, rgb_every=100, state_every=10,
       fused="world")


def test_beam_heavy_actions(clean_up_pack):
  # NOOP FWD BACK LEFT RIGHT TURN_L TURN_R ZAP CLEAN: half the actions fire
  w = [1, 2, 1, 1, 1, 1, 1, 4, 4]
  _run(clean_up_pack, n=16, steps=400, seed=7, weights=w, rgb_every=20)


def test_movement_heavy_many_worlds(clean_up_pack):
  w = [0, 6, 2, 2, 2, 2, 2, 1, 1]
  _run(clean_up_pack, n=64, steps=150, seed=11, weights=w, rgb_every=50,
       state_every=5)


def test_apples_grow_and_get_eaten(clean_up_pack):
  """AppleGrow / Edible / Taste (clean_up/components.lua:64-80,390-455): needs a
  pack whose growth thresholds let apples appear under random play."""
  from meltingpot_amd import engine as E
  pack = util.fertile_clean_up(clean_up_pack)
  w = [0, 8, 2, 3, 3, 2, 2, 1, 2]  # mostly walking
  _run(pack, n=16, steps=300, seed=21, weights=w, rgb_every=25)
  eng = _engine(pack, 64)
  eng.reset()
  import torch
  rng = np.random.default_rng(2)
  acts = util.random_actions(rng, 200, 64, eng.P, eng.num_actions, w)
  total = 0.0
  for s in range(200):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    total += float(eng.observe(E.OBS_COLLECTIVE_REWARD).sum())
  assert total > 100, total  # apples were eaten: the rule path is really exercised
  assert eng.counters()["reward_sum_x1024"] == int(total * 1024)
  eng.close()


def test_episode_end_and_auto_reset(clean_up_pack):
  """maxEpisodeLengthFrames cap (api_factory.lua:107-110) and the
  rebuild-with-seed+1 reset convention (builder.py:177-181)."""
  import torch
  from meltingpot_amd import engine as E
  pack = util.patch_pack(clean_up_pack, MAXFRAMES=25)
  n = 4
  eng = _engine(pack, n, auto_reset=True)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(3)
  acts = util.random_actions(rng, 80, n, eng.P, eng.num_actions)
  dacts = torch.from_numpy(acts).to(eng.device)
  for s in range(80):
    eng.step(dacts[s])
    st = eng.observe(E.OBS_STEP_TYPE).cpu().numpy()
    for w, o in enumerate(oracles):
      if o.done:          # dm_env: step after LAST restarts the episode
        o.reset()
        assert st[w] == 0
      else:
        cont = o.step(acts[s, w])
        assert st[w] == (1 if cont else 2), (s, w)
    _compare_state(eng, oracles, f"step {s + 1}")
    _compare_scalars(eng, oracles, f"step {s + 1}")
  _compare_rgb(eng, oracles, "end")
  c = eng.counters()
  assert c["episodes"] == n * 4 and c["world_steps"] == n * (80 - 3)
  eng.close()


def test_frozen_without_auto_reset_and_masked_reset(clean_up_pack):
  import torch
  from meltingpot_amd import engine as E
  pack = util.patch_pack(clean_up_pack, MAXFRAMES=5)
  n = 3
  eng = _engine(pack, n, auto_reset=False)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  acts = np.ones((n, eng.P), np.int32)
  for _ in range(8):
    eng.step(torch.from_numpy(acts).to(eng.device))
    for o in oracles:
      o.step(acts[0])
  _compare_state(eng, oracles, "frozen")
  assert (eng.observe(E.OBS_STEP_TYPE).cpu().numpy() == 2).all()
  eng.reset(mask=[0, 1, 0])
  oracles[1].reset()
  _compare_state(eng, oracles, "masked reset")
  eng.close()


def test_snapshot_restore_and_sharding_invariance(clean_up_pack):
  """Worlds are seeded from their global index: an engine owning worlds
  [4, 8) reproduces worlds 4..7 of an engine owning [0, 8)."""
  import torch
  from meltingpot_amd import engine as E
  a = _engine(clean_up_pack, 8)
  b = _engine(clean_up_pack, 4, world_offset=4)
  a.reset(); b.reset()
  rng = np.random.default_rng(5)
  acts = util.random_actions(rng, 40, 8, a.P, a.num_actions)
  snap = None
  for s in range(40):
    a.step(torch.from_numpy(acts[s]).to(a.device))
    b.step(torch.from_numpy(acts[s, 4:]).to(b.device))
    if s == 19:
      snap = a.snapshot()
  ga, aa, la = a.dump()
  gb, ab, lb = b.dump()
  assert np.array_equal(ga[4:], gb) and np.array_equal(aa[4:], ab)
  assert np.array_equal(la[4:], lb)
  ra = a.observe(E.OBS_RGB).cpu().numpy()
  rb = b.observe(E.OBS_RGB).cpu().numpy()
  assert np.array_equal(ra[4:], rb)
  a.restore(snap)
  for s in range(20, 40):
    a.step(torch.from_numpy(acts[s]).to(a.device))
  g2, a2, l2 = a.dump()
  assert np.array_equal(g2, ga) and np.array_equal(a2, aa) and np.array_equal(l2, la)
  a.close(); b.close()


def test_bound_outputs_and_host_actions(clean_up_pack):
  import torch
  from meltingpot_amd import engine as E
  n = 5
  eng = _engine(clean_up_pack, n)
  rgb = eng.bind(E.OBS_RGB)
  wrgb = eng.bind(E.OBS_WORLD_RGB)
  rew = eng.bind(E.OBS_REWARD)
  eng.reset()
  oracles = util.make_oracles(clean_up_pack, n)
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(9)
  for s in range(30):
    acts = util.random_actions(rng, 1, n, eng.P, eng.num_actions)[0]
    eng.step(acts)  # host path: validated + uploaded by mp_step_host
    for w, o in enumerate(oracles):
      o.step(acts[w])
  torch.cuda.synchronize()
  for w, o in enumerate(oracles):
    assert np.array_equal(wrgb[w].cpu().numpy(), o.render_world())
    assert np.array_equal(rew[w].cpu().numpy(), o.rewards())
    for p in range(o.P):
      assert np.array_equal(rgb[w, p].cpu().numpy(), o.render_agent(p))
  with pytest.raises(ValueError):
    bad = np.zeros((n, eng.P), np.int32)
    bad[2, 3] = eng.num_actions
    eng.step(bad)
  eng.close()


# ---------------------------------------------------------------- commons_harvest__open
# (BASELINE.json configs[2]: 16 players; 8 actions: NOOP FWD RIGHT BACK LEFT
# TURN_L TURN_R ZAP, commons_harvest__open.py:264-273)


def test_commons_reset_and_rollout(commons_pack):
  _run(commons_pack, n=8, steps=120, seed=1, rgb_every=10)


def test_commons_1000_fixed_seed_steps(commons_pack):
  """north_star: "bit-exact parity vs reference on 1000 fixed-seed steps" — 64 worlds
  like clean_up's, state and scalars after every step, both views every 100."""
  _run(commons_pack, n=64, steps=1000, seed=1234, rgb_every=100)


def test_commons_harvest_heavy(commons_pack):
  """Walking-heavy play eats the orchard down: exercises DensityRegrow's wait
  states, dessication and regrowth (commons_harvest/components.lua:71-240)."""
  from meltingpot_amd import engine as E
  w = [0, 8, 3, 2, 3, 2, 2, 1]
  _run(commons_pack, n=16, steps=500, seed=5, weights=w, rgb_every=50)
  eng = _engine(commons_pack, 32)
  eng.reset()
  import torch
  rng = np.random.default_rng(3)
  acts = util.random_actions(rng, 300, 32, eng.P, eng.num_actions, w)
  total = 0.0
  for s in range(300):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    total += float(eng.observe(E.OBS_COLLECTIVE_REWARD).sum())
  _, _, glob = eng.dump()
  assert total > 500 and (glob[:, 3] < 64).all()  # apples eaten, orchards depleted
  eng.close()


def test_commons_zap_heavy(commons_pack):
  w = [1, 3, 1, 1, 1, 2, 2, 8]
  _run(commons_pack, n=16, steps=300, seed=8, weights=w, rgb_every=30)


def test_commons_episode_end_and_auto_reset(commons_pack):
  import torch
  from meltingpot_amd import engine as E
  pack = util.patch_pack(commons_pack, MAXFRAMES=20)
  n = 4
  eng = _engine(pack, n, auto_reset=True)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(3)
  acts = util.random_actions(rng, 70, n, eng.P, eng.num_actions)
  for s in range(70):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      if o.done:
        o.reset()
      else:
        o.step(acts[s, w])
    _compare_state(eng, oracles, f"step {s + 1}")
    _compare_scalars(eng, oracles, f"step {s + 1}")
  _compare_rgb(eng, oracles, "end")
  eng.close()


def test_commons_closed_variant(commons_closed_pack):
  """commons_harvest__closed: same Lua level, walled-orchard map, 7 players —
  runs on the commons_harvest kernels with no new rule code."""
  w = [0, 8, 3, 2, 3, 2, 2, 2]
  _run(commons_closed_pack, n=8, steps=400, seed=4, weights=w, rgb_every=40)


def test_commons_partnership_variant(commons_partnership_pack):
  """commons_harvest__partnership: two spawn groups (2 players inside the
  orchard), hidden role-based reward tiles that are inert for the default roles
  (component_library.lua:1097-1133) — again no new rule code."""
  w = [0, 8, 3, 2, 3, 2, 2, 2]
  _run(commons_partnership_pack, n=8, steps=400, seed=5, weights=w, rgb_every=40)


# ---------------------------------------------------------------- coins
# (a level with its own rule code — Coin, ChoiceCoinRegrow,
# PartnerTracker — and no Zapper; 2 players, 7 actions)


@pytest.mark.parametrize("unfused", [True, None])
def test_coins_with_a_bound_view(coins_pack, unfused):
  """The per-agent view of two players is small (46 KB a world): since round 3 the
  engine's own choice (MpConfig.unfused = 0, None here) is the fused launch for it
  too, in batches of 8; unfused=True = one launch for the rules, one for the view."""
  _run(coins_pack, n=8, steps=200, seed=33, weights=[0, 8, 2, 2, 2, 1, 1], rgb_every=20,
       fused="agents", unfused=unfused)


def test_coins_rollouts(coins_pack):
  """Episodes end early here (StochasticIntervalEpisodeEnding from frame 300,
  coins.py:120-127), so the rollout follows the auto-reset: state, rewards, the
  partner-mismatch observation and both views, through several episodes."""
  import torch
  n, steps = 16, 1500
  eng = _engine(coins_pack, n, auto_reset=True)
  oracles = util.make_oracles(coins_pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  _compare_state(eng, oracles, "reset")
  _compare_rgb(eng, oracles, "reset")
  rng = np.random.default_rng(32)
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions, [0, 8, 2, 2, 2, 1, 1])
  restarts, collected = 0, 0.0
  for s in range(steps):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    fresh = []
    for w, o in enumerate(oracles):
      if o.done:
        o.reset(); restarts += 1; fresh.append(w)
      else:
        o.step(acts[s, w])
        collected += float(np.maximum(o.rewards(), 0).sum())
    _compare_state(eng, oracles, f"step {s + 1}")
    if not fresh:
      _compare_scalars(eng, oracles, f"step {s + 1}")
    if (s + 1) % 100 == 0:
      _compare_rgb(eng, oracles, f"step {s + 1}")
  assert restarts >= 3 and collected > 50
  # every world has the map size its build drew (coins.py:45-82): more than one here
  sizes = set()
  for g in eng.dump()[0]:
    ys, xs = np.where((g != 0).any(axis=0))
    sizes.add((int(xs.max()), int(ys.max())))
  assert len(sizes) >= 4, sizes
  # ... and the two colours its build drew (coins.py:500-514)
  from meltingpot_amd import pack
  coin = pack.loads(coins_pack)["co_colour_coin"].tolist()
  pairs = {tuple(sorted(coin.index(int(v)) for v in np.unique(g) if int(v) in coin))
           for g in eng.dump()[0]}
  assert len(pairs) >= 4, pairs
  eng.close()


# ---------------------------------------------------------------- territory__rooms
# (BASELINE.json configs[3]: 9 players, TORUS; 9 actions: NOOP FWD BACK LEFT RIGHT
# TURN_L TURN_R ZAP CLAIM, territory.py:592-602)


@pytest.mark.parametrize("unfused", [None, False, True])
def test_territory_reset_and_rollout(territory_pack, unfused):
  """None: the engine's choice of launches (the fused one: MpConfig.unfused);
  False / True: one fused launch / one for the rules and one per view."""
  _run(territory_pack, n=8, steps=150, seed=1, rgb_every=10, unfused=unfused)


def test_territory_1000_fixed_seed_steps(territory_pack):
  _run(territory_pack, n=64, steps=1000, seed=1234, rgb_every=100)


def test_territory_as_benchmarked(territory_pack):
  """BASELINE.json configs[3] exactly as bench.py measures it: 8192 worlds, 9
  players, per-agent RGB bound (one fused launch per step), HALF of the actions the
  two beam actions (`--beam-skew 0.5`), 500 steps so that the
  timed phase of the episode (steps 300 - 500: most of the map claimed, painted and
  partly destroyed) is reached.  The oracle replays 8 blocks of 64 CONSECUTIVE worlds
  spread over the batch (512 of the 8192 worlds: replaying all of them for 500 steps
  would take minutes of the GPU box's time) — state, hidden rule variables, rewards and
  events of step 500 bit-exact, the bound view of 4 worlds per block after steps 300,
  400 and 500; the counters cover all 8192."""
  import torch
  from meltingpot_amd import engine as E
  n, steps, looks = 8192, 500, (300, 400, 500)
  eng = _engine(territory_pack, n)
  view = eng.bind(E.OBS_RGB)
  assert eng.fused
  eng.reset()
  gen = torch.Generator(device=eng.device)
  gen.manual_seed(1234)
  T = 250   # (an action ring, as in bench.py)
  acts = torch.randint(0, eng.num_actions, (T, n, eng.P), generator=gen, device=eng.device,
                       dtype=torch.int32)
  beam = torch.randint(eng.num_actions - 2, eng.num_actions, (T, n, eng.P), generator=gen,
                       device=eng.device, dtype=torch.int32)
  pick = torch.rand((T, n, eng.P), generator=gen, device=eng.device) < 0.5
  acts = torch.where(pick, beam, acts)
  blocks = [b * 1024 + 37 * b for b in range(8)]          # 64 consecutive worlds each
  blocks[-1] = n - 64                                      # ... the last 64 among them
  sample = [w0 + k for w0 in blocks for k in (0, 21, 42, 63)]
  sel = torch.tensor(sample, device=eng.device)
  seen = {}
  for s in range(steps):
    eng.step(acts[s % T])
    if s + 1 in looks:
      seen[s + 1] = view[sel].cpu().numpy()
  c = eng.counters()
  assert c["world_steps"] == n * steps and c["bad_actions"] == 0 and c["zaps"] > 0
  host = acts.cpu().numpy()
  host = np.concatenate([host, host], 0)[:steps]           # the ring, unrolled
  grid, avat, glob = eng.dump()
  rew = eng.observe(E.OBS_REWARD).cpu().numpy()
  ev = eng.observe(E.OBS_EVENTS).cpu().numpy()
  where = {w: i for i, w in enumerate(sample)}
  replayed = 0
  for w0 in blocks:
    for w, og, oa, ogl, orew, oev, views in util.replay_parallel(
        territory_pack, host[:, w0:w0 + 64], looks=looks, sample=sample, world_view=False,
        offset=w0):
      assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa), w
      assert np.array_equal(glob[w], ogl) and np.array_equal(rew[w], orew), w
      got = sorted(tuple(int(v) for v in r[:3]) for r in ev[w, 1:1 + int(ev[w, 0, 0])])
      assert got == oev, w
      if w in where:
        assert sorted(views) == sorted(looks)
        for step, want in views.items():
          assert np.array_equal(seen[step][where[w]], want), (w, step)
      replayed += 1
  assert replayed == 512
  assert not eng.fault_words()[:6].any()
  eng.close()


@pytest.mark.parametrize("unfused", [None, False])
def test_territory_beam_heavy(territory_pack, unfused):
  """actions skewed towards FIRE_ZAP / FIRE_CLAIM — resource
  damage, destruction, self repair, sanctions (freeze, removal), claims."""
  w = [1, 4, 1, 1, 1, 2, 2, 6, 6]
  _run(territory_pack, n=16, steps=500, seed=6, weights=w, rgb_every=50, unfused=unfused)


def test_territory_movement_heavy(territory_pack):
  w = [0, 8, 2, 2, 2, 3, 3, 1, 1]
  _run(territory_pack, n=32, steps=300, seed=9, weights=w, rgb_every=60, state_every=3)


def test_territory_open_map(territory_open_pack):
  """territory__open: the same Lua level on the 23 x 39 BOUNDED map
  (territory__open.py:45-70) — a different world size, topology and view
  clipping through the same kernels."""
  _run(territory_open_pack, n=6, steps=300, seed=21, rgb_every=60)
  w = [1, 4, 1, 1, 1, 2, 2, 6, 6]
  _run(territory_open_pack, n=6, steps=200, seed=22, weights=w, rgb_every=50)


def test_territory_inside_out_choice_maps(territory_inside_out_pack):
  """territory__inside_out: 'A' / 'B' / 'Q' map characters are `choice` prefabs
  (prefab_utils.lua:101-103) — optional resources and spawn points drawn once per
  episode.  Short episodes with auto-reset walk through many different maps:
  state (incl. which resources exist), rewards, spawn cells and both views."""
  import torch
  pack = util.patch_pack(territory_inside_out_pack, MAXFRAMES=40)
  n, steps = 12, 260
  eng = _engine(pack, n, auto_reset=True)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  _compare_state(eng, oracles, "reset")
  _compare_rgb(eng, oracles, "reset")
  maps = {tuple(o.dump()[0][3].ravel().tolist()) for o in oracles}   # resource texture plane
  assert len(maps) == n, "every world should have drawn its own map"
  rng = np.random.default_rng(41)
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions, [1, 4, 1, 1, 1, 2, 2, 4, 4])
  for s in range(steps):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    fresh = False
    for w, o in enumerate(oracles):
      if o.done:
        o.reset(); fresh = True
      else:
        o.step(acts[s, w])
    _compare_state(eng, oracles, f"step {s + 1}")
    if not fresh:
      _compare_scalars(eng, oracles, f"step {s + 1}")
    if fresh or (s + 1) % 50 == 0:
      _compare_rgb(eng, oracles, f"step {s + 1}")
  eng.close()
  # and a long episode on the unpatched pack
  _run(territory_inside_out_pack, n=6, steps=400, seed=42, weights=[1, 4, 1, 1, 1, 2, 2, 6, 6],
       rgb_every=80)


def test_territory_episode_end_and_auto_reset(territory_pack):
  import torch
  pack = util.patch_pack(territory_pack, MAXFRAMES=30)
  n = 4
  eng = _engine(pack, n, auto_reset=True)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(3)
  acts = util.random_actions(rng, 100, n, eng.P, eng.num_actions)
  for s in range(100):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      if o.done:
        o.reset()
      else:
        o.step(acts[s, w])
    _compare_state(eng, oracles, f"step {s + 1}")
    _compare_scalars(eng, oracles, f"step {s + 1}")
  _compare_rgb(eng, oracles, "end")
  eng.close()


# ---------------------------------------------------------------- renderer launch geometry


@pytest.mark.parametrize("n,batch,waves,feeders", [
    (1, 0, 0, 0), (3, 0, 0, 0), (37, 8, 5, 2), (37, 3, 16, 3), (130, 4, 8, 1), (130, 2, 2, 1),
    (1030, 0, 0, 0), (1100, 5, 16, 5)])
def test_frame_geometry_edge_cases(clean_up_pack, commons_pack, n, batch, waves, feeders):
  """The persistent frame kernel: workgroups own contiguous ranges of worlds and
  walk them in batches of `batch` through two LDS buffers, `feeders` of the
  `waves` waves feeding.  World counts that do not divide, partial last batches
  and workgroups, single worlds, more than one range per CU (> 1024 worlds) and
  the planner's own choices must all step and render every world bit-exactly —
  in the fused launch (the bound view) and in the render-only one."""
  import torch
  from meltingpot_amd import engine as E
  dev = {"batch_worlds": batch, "waves": waves, "feeders": feeders} if batch else None
  for pack, bound_kind in ((clean_up_pack, E.OBS_WORLD_RGB), (commons_pack, E.OBS_RGB)):
    eng = _engine(pack, n, dev=dev)   # MpConfig.dev: test-only plan overrides
    bound = eng.bind(bound_kind)
    eng.reset()
    rng = np.random.default_rng(n)
    acts = util.random_actions(rng, 6, n, eng.P, eng.num_actions)
    for s in range(6):
      eng.step(torch.from_numpy(acts[s]).to(eng.device))
    rgb = (bound if bound_kind == E.OBS_RGB else eng.observe(E.OBS_RGB)).cpu().numpy()
    wrgb = (bound if bound_kind == E.OBS_WORLD_RGB else eng.observe(E.OBS_WORLD_RGB)).cpu().numpy()
    grid, avat, glob = eng.dump()
    sample = sorted(set([0, n - 1, n // 2] + list(rng.integers(0, n, 5))))
    for w in sample:
      o = util.make_oracles(pack, 1, offset=int(w))[0]
      o.reset()
      for s in range(6):
        o.step(acts[s, w])
      og, oa, ogl = o.dump()
      assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa), (n, w)
      assert np.array_equal(glob[w], ogl), (n, w)
      assert np.array_equal(wrgb[w], o.render_world()), (n, w)
      for p in range(o.P):
        assert np.array_equal(rgb[w, p], o.render_agent(p)), (n, w, p)
    eng.close()


@pytest.mark.parametrize("which,n,dev", [
    ("clean_up", 150, None),                                        # the product's plan
    ("clean_up", 150, {"batch_worlds": 1, "ring_batches": 6, "static_pct": 50, "max_groups": 5}),
    ("clean_up", 40, {"waves": 2, "max_groups": 3}),               # (raised to a feeder + a wave per view)
    ("clean_up", 70, {"waves": 5, "feeders": 2, "world_waves": 2, "batch_worlds": 2, "max_groups": 3}),
    ("commons", 90, {"static_pct": 50, "max_groups": 4, "store_sc1": 1}),
    ("territory", 80, {"batch_worlds": 1, "ring_batches": 6, "feeders": 3, "static_pct": 40, "max_groups": 3}),
])
def test_both_views_in_one_launch(clean_up_pack, commons_pack, territory_pack, which, n, dev):
  """`substrate.build(..., num_worlds=N)` binds per-agent RGB AND WORLD.RGB: one
  k_frame<..., 2> launch steps the worlds and draws both from the same LDS-resident
  records (the renderer waves split between the views).  Under the product's plan and
  under forced ones — single-world batches through a deep ring, half of them pooled
  behind the device-wide claim counter, minimal wave counts, sc1 stores — state, scalars
  and every pixel of both views against the oracle, worlds restarting on the way."""
  pack = {"clean_up": clean_up_pack, "commons": commons_pack, "territory": territory_pack}[which]
  _run(pack, n=n, steps=40, seed=n, rgb_every=8, fused="both", unfused=False, dev=dev)
  short = util.patch_pack(pack, MAXFRAMES=11)
  import torch
  from meltingpot_amd import engine as E
  eng = _engine(short, n, auto_reset=True, unfused=False, dev=dev)
  eng.bind(E.OBS_RGB); eng.bind(E.OBS_WORLD_RGB)
  assert eng.fused
  oracles = util.make_oracles(short, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(n + 1)
  acts = util.random_actions(rng, 30, n, eng.P, eng.num_actions)
  for s in range(30):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      if o.done:
        o.reset()
      else:
        o.step(acts[s, w])
    if s % 5 == 4:
      _compare_state(eng, oracles, f"step {s + 1}")
      _compare_rgb(eng, oracles, f"step {s + 1}")
  assert not eng.fault_words()[:6].any()
  eng.close()


@pytest.mark.parametrize("which,view,n,dev", [
    # head=1: the older road (tables waited for in the prologue, every record loaded in the loop)
    ("clean_up", "world", 70, {"head": 1}),
    ("commons", "agents", 40, {"head": 1, "static_pct": 50, "max_groups": 3}),
    ("matrix", "both", 40, {"head": 1}),
    # head=2, the product's (tables, first record and its action ids by LDS DMA): fewer
    # worlds than a workgroup has feeders (a feeder with nothing to request), feeders whose
    # first slot lies in the ring's second batch, a first batch that is a POOLED one for
    # most workgroups (those feeders take the older road), records of every size
    ("clean_up", "world", 1, {"head": 2}),
    ("clean_up", "both", 3, {"head": 2}),
    ("clean_up", "world", 70, {"head": 2, "batch_worlds": 1, "ring_batches": 8}),
    ("clean_up", "agents", 300, {"head": 2, "batch_worlds": 1, "ring_batches": 8, "static_pct": 25, "max_groups": 7}),
    ("commons", "agents", 40, {"head": 2, "feeders": 3}),
    ("territory", "agents", 40, {"head": 2, "batch_worlds": 2, "ring_batches": 3, "feeders": 3}),
    ("coins", "both", 130, {"head": 2}),
    ("matrix", "both", 40, {"head": 2, "static_pct": 50, "max_groups": 3}),
])
def test_both_roads_into_a_stepping_launch(clean_up_pack, commons_pack, territory_pack, coins_pack,
                                           which, view, n, dev):
  """How a stepping launch starts (FramePlan::head, csrc/frame.hip; profiles/r04_head.md):
  the product requests a feeder's tables, its first world's record and that world's
  action ids by global -> LDS DMA before anything else and waits for them at its first
  step; the older road stays selectable.  Both, under plans that put a feeder's first
  world in every position, against the oracle: state, scalars, pixels, with host-side
  action arrays too (the DMA then reads pinned host memory)."""
  from meltingpot_amd import engine as E
  pack = {"clean_up": clean_up_pack, "commons": commons_pack, "territory": territory_pack,
          "coins": coins_pack,
          "matrix": E.load_pack("prisoners_dilemma_in_the_matrix__arena")}[which]
  _run(pack, n=n, steps=14, seed=n + 3, rgb_every=7, fused=view, unfused=False, dev=dev)
  eng = _engine(pack, n, unfused=False, dev=dev)
  eng.bind(E.OBS_WORLD_RGB if view == "world" else E.OBS_RGB)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(n)
  acts = util.random_actions(rng, 6, n, eng.P, eng.num_actions)
  for s in range(6):
    eng.step(acts[s])                      # a host array: mp_step_host
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
  _compare_state(eng, oracles, "host actions")
  _compare_rgb(eng, oracles, "host actions")
  assert not eng.fault_words()[:6].any()
  eng.close()


@pytest.mark.parametrize("dev", [
    {"no_composite_cache": 1},                        # every overlay composited on the fly
    {"no_composite_cache": 1, "scratch_cells": 2},    # mostly the direct-store path
    {"max_composites": 7, "scratch_cells": 24},
])
def test_render_paths_agree_with_the_oracle(clean_up_pack, territory_pack, dev):
  """The renderer's shortcuts — the composite cache of static stacks, the LDS
  staging of composited cells, the direct-store path for crowded passes — are
  all bit-exact: switch them off / squeeze them and compare with the oracle."""
  import torch
  for pack in (clean_up_pack, territory_pack):
    n = 6
    eng = _engine(pack, n, dev=dev)
    oracles = util.make_oracles(pack, n)
    eng.reset()
    for o in oracles:
      o.reset()
    rng = np.random.default_rng(11)
    acts = util.random_actions(rng, 40, n, eng.P, eng.num_actions)
    for s in range(40):
      eng.step(torch.from_numpy(acts[s]).to(eng.device))
      for w, o in enumerate(oracles):
        o.step(acts[s, w])
      if s % 13 == 0:
        _compare_rgb(eng, oracles, f"step {s + 1} {dev}")
    _compare_rgb(eng, oracles, f"end {dev}")
    eng.close()


@pytest.mark.parametrize("dev", [
    None,                                             # hinted stacks from the cache
    {"scratch_cells": 2},                             # ... the rest stored directly
    {"max_composites": 30},                           # the cache holds the pairs only
    {"no_composite_cache": 1, "scratch_cells": 8},    # dozens of composited cells a pass
])
@pytest.mark.parametrize("which", ["rooms", "open"])
def test_territory_late_in_an_episode(territory_pack, territory_open_pack, which, dev):
  """Late in an episode most resources are claimed and pay: texture + wet paint +
  dry paint of one player (territory.py:356-507), the stack `composite_hints` names.
  The drawing of such a map — crowded passes included, cached or not — against the
  oracle, both views, the per-agent one by the launch that steps."""
  pack = territory_pack if which == "rooms" else territory_open_pack
  # NOOP FWD BACK LEFT RIGHT TURN_L TURN_R ZAP CLAIM: walk and claim
  _run(pack, n=4, steps=330, seed=21, weights=[1, 6, 1, 2, 2, 2, 2, 1, 8], rgb_every=110,
       state_every=110, fused="agents", dev=dev)


@pytest.mark.parametrize("which,weights", [
    ("clean_up", [1, 4, 1, 1, 1, 2, 2, 6, 6]),
    ("commons", [1, 6, 1, 1, 1, 2, 2, 6]),
    ("territory", [1, 4, 1, 1, 1, 2, 2, 6, 6]),
    ("coins", [0, 8, 2, 2, 2, 1, 1]),
])
def test_events_channel(clean_up_pack, commons_pack, territory_pack, coins_pack, which, weights):
  """env.events() (wrappers/base.py:72-74): every `events:add` of the hot path
  — zap, edible_consumed, player_cleaned, claimed/destroyed_resource, the
  sanctioning events, AvatarStarted — as a multiset per world-step, compared
  with the oracle's log after reset and after every step."""
  import torch
  from meltingpot_amd import engine as E
  pack = {"clean_up": clean_up_pack, "commons": commons_pack, "territory": territory_pack,
          "coins": coins_pack}[which]
  n, steps = 8, 250
  eng = _engine(pack, n)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()

  def check(tag):
    rows = eng.observe(E.OBS_EVENTS).cpu().numpy()
    seen = set()
    for w, o in enumerate(oracles):
      cnt = int(rows[w, 0, 0])
      assert rows[w, 0, 1] == 0, "event rows dropped"
      got = sorted(tuple(int(v) for v in r[:3]) for r in rows[w, 1:1 + cnt])
      assert got == o.events(), (tag, w, got, o.events())
      seen.update(t for t, _, _ in got)
    return seen

  seen = check("reset")
  assert E.EVENT_TYPES[9][0] == "AvatarStarted" and 9 in seen
  rng = np.random.default_rng(17)
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions, weights=weights)
  for s in range(steps):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
    seen |= check(f"step {s + 1}")
  want = {"clean_up": {1, 3}, "commons": {1, 2}, "territory": {1, 4, 5, 6, 8},
          "coins": {10}}[which]
  assert want <= seen, (want, seen)
  # the decoded form of the reference API
  names = {name for name, _ in eng.events(0)} | {E.EVENT_TYPES[t][0] for t in seen}
  assert all(isinstance(x, str) for x in names)
  eng.close()


@pytest.mark.parametrize("which,n,world,bound", [
    ("clean_up", 4096, True, True),      # BASELINE.json configs[1]
    ("clean_up", 4096, True, False),
    ("commons", 4096, False, True),      # configs[2]
    ("commons", 4096, False, False),
    ("territory", 8192, False, True),    # configs[3]
    ("territory", 8192, False, False),
    # what `substrate.build("clean_up", ..., num_worlds=4096)` binds and bench.py's
    # `substrate_api` times: BOTH views before the first step, one k_frame<..., 2> launch
    ("clean_up", 4096, "both", True),
    # the levels of round 5 at the same size
    ("coop_mining", 4096, "both", True),
    ("gift_refinements", 4096, False, True),
    ("collaborative_cooking__crowded", 4096, "both", True),
    ("collaborative_cooking__cramped", 4096, False, True),
    ("externality_mushrooms__dense", 4096, "both", True),
])
def test_full_size_properties(clean_up_pack, commons_pack, territory_pack, which, n, world, bound):
  """BASELINE.json's full batch sizes, in the launch form bench.py times
  (`bound`: the benchmarked view is bound BEFORE the first step, so all 64 steps
  are the fused k_frame<...Tables> with 4+ batches per workgroup through the
  two-buffer ring) and with no view bound (stand-alone step kernel + render-only
  k_frame): EVERY world replayed by the oracle for 64 steps — state, hidden rule
  variables, rewards and events bit-exact — and the benchmarked observation
  compared on 256 sampled worlds (first, last, middle, random) after steps 17, 41
  and 64; the counters against the exact number of world-steps; the reward
  counter against the sum of the reward tensor; and a second engine started at a
  world offset reproducing the first one's tail (sharding invariance at full
  size)."""
  import torch
  from meltingpot_amd import engine as E
  pack = {"clean_up": clean_up_pack, "commons": commons_pack, "territory": territory_pack}.get(which)
  if pack is None:
    pack = E.load_pack(which)
  steps, looks = 64, (17, 41, 64)
  both = world == "both"
  kind = E.OBS_WORLD_RGB if world else E.OBS_RGB
  eng = _engine(pack, n)
  view = eng.bind(kind) if bound else None
  agents_view = eng.bind(E.OBS_RGB) if both else None
  if bound:
    assert eng.fused   # one launch per step: rules + this view (or both views)
  eng.reset()
  gen = torch.Generator(device=eng.device)
  gen.manual_seed(99)
  acts = torch.randint(0, eng.num_actions, (steps, n, eng.P), generator=gen,
                       device=eng.device, dtype=torch.int32)
  rng = np.random.default_rng(5)
  rgb_sample = sorted({0, n - 1, n // 2, *map(int, rng.integers(0, n, 253))})
  pick = torch.tensor(rgb_sample, device=eng.device)
  total = torch.zeros((), dtype=torch.float64, device=eng.device)
  total_fx = torch.zeros((), dtype=torch.int64, device=eng.device)
  seen = {}
  for s in range(steps):
    eng.step(acts[s])
    total += eng.observe(E.OBS_REWARD).sum()
    # (the counter adds every world-step's collective reward in whole 1/1024 units:
    # externality_mushrooms pays fifths and quarters)
    total_fx += torch.trunc(eng.observe(E.OBS_COLLECTIVE_REWARD) * 1024.0).to(torch.int64).sum()
    if s + 1 in looks:
      seen[s + 1] = (view if bound else eng.observe(kind))[pick].cpu().numpy()
      if both:
        seen[s + 1] = (seen[s + 1], agents_view[pick].cpu().numpy())
  rgb = view if bound else eng.observe(kind)
  c = eng.counters()
  assert c["world_steps"] == n * steps and c["agent_steps"] == n * steps * eng.P
  assert c["episodes"] == n and c["bad_actions"] == 0
  assert c["reward_sum_x1024"] == int(total_fx)
  assert abs(int(total_fx) - float(total) * 1024) <= n * steps   # (and the reward tensor's own sum)
  host_acts = acts.cpu().numpy()
  grid, avat, glob = eng.dump()
  rew = eng.observe(E.OBS_REWARD).cpu().numpy()
  ev = eng.observe(E.OBS_EVENTS).cpu().numpy()
  where = {w: i for i, w in enumerate(rgb_sample)}
  replayed = 0
  for w, og, oa, ogl, orew, oev, views in util.replay_parallel(
      pack, host_acts, looks=looks, sample=rgb_sample, world_view=world):
    assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa), w
    assert np.array_equal(glob[w], ogl) and np.array_equal(rew[w], orew), w
    got = sorted(tuple(int(v) for v in r[:3]) for r in ev[w, 1:1 + int(ev[w, 0, 0])])
    assert got == oev, w
    if w in where:
      assert sorted(views) == sorted(looks)
      for step, want in views.items():
        if both:
          assert np.array_equal(seen[step][0][where[w]], want[0]), (w, step, "WORLD.RGB")
          assert np.array_equal(seen[step][1][where[w]], want[1]), (w, step, "RGB")
        else:
          assert np.array_equal(seen[step][where[w]], want), (w, step)
    replayed += 1
  assert replayed == n
  # the last 64 worlds again, as their own shard
  tail = _engine(pack, 64, world_offset=n - 64)
  tail_view = tail.bind(kind) if bound else None
  tail_agents = tail.bind(E.OBS_RGB) if both else None
  tail.reset()
  for s in range(steps):
    tail.step(acts[s, n - 64:].contiguous())
  assert torch.equal(tail_view if bound else tail.observe(kind), rgb[n - 64:])
  if both:
    assert torch.equal(tail_agents, agents_view[n - 64:])
  tg, ta, tgl = tail.dump()
  assert np.array_equal(tg, grid[n - 64:]) and np.array_equal(ta, avat[n - 64:])
  tail.close()
  eng.close()


@pytest.mark.parametrize("which,n,groups,auto_reset", [
    ("clean_up", 150, 5, False),     # 30 worlds a workgroup: 8 batches of 4 through 2 buffers
    ("commons", 100, 4, False),      # 25 (27 rounded to whole batches of 3): 9 batches
    ("territory", 100, 4, False),
    ("coins", 120, 3, False),
    ("clean_up", 90, 2, True),       # ... with worlds restarting inside the feeders
    ("coins", 90, 2, True),
])
def test_fused_ring_recycles_buffers(clean_up_pack, commons_pack, territory_pack, coins_pack,
                                     which, n, groups, auto_reset):
  """The fused launch with MANY batches per workgroup, for every step
  instantiation of k_frame: a feeder may refill buffer k & 1 only once all passes
  of batch k - 2 are drawn (frame.hip: buffer_free), and it is stepping worlds
  while it waits its turn.  `max_groups` (MpConfig.dev, test-only) caps the
  workgroups so that a small batch of worlds walks the two-buffer ring 8+ times
  per launch; every world and every pixel of the bound view is compared with the
  oracle after every few steps (short episodes + auto-reset in the last cases,
  so resets run inside the ring as well)."""
  import torch
  from meltingpot_amd import engine as E
  pack = {"clean_up": clean_up_pack, "commons": commons_pack, "territory": territory_pack,
          "coins": coins_pack}[which]
  if auto_reset:
    pack = util.patch_pack(pack, MAXFRAMES=9)
  steps = 30
  eng = _engine(pack, n, auto_reset=auto_reset, unfused=False, dev={"max_groups": groups})
  kind = E.OBS_WORLD_RGB if which == "clean_up" else E.OBS_RGB
  eng.bind(kind)
  assert eng.fused
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  _compare_state(eng, oracles, "reset")
  _compare_rgb(eng, oracles, "reset")
  rng = np.random.default_rng(n)
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions)
  restarts = 0
  for s in range(steps):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      if o.done and auto_reset:
        o.reset(); restarts += 1
      else:
        o.step(acts[s, w])
    if s % 4 == 3 or s == steps - 1:
      _compare_state(eng, oracles, f"step {s + 1}")
      _compare_rgb(eng, oracles, f"step {s + 1}")
  assert restarts >= (2 * n if auto_reset else 0)
  assert not eng.fault_words()[:6].any()
  eng.close()


@pytest.mark.parametrize("which", ["clean_up", "commons", "territory"])
def test_natural_episode_ends(clean_up_pack, commons_pack, territory_pack, which):
  """StochasticIntervalEpisodeEnding (component_library.lua:907-948) on the
  unpatched packs: 2200 steps with auto-reset, so worlds end by the interval
  draw after frame 1000 and restart with seed + 1 (tests/tools/soak.py runs
  the same on every pack)."""
  import torch
  from meltingpot_amd import engine as E
  pack = {"clean_up": clean_up_pack, "commons": commons_pack, "territory": territory_pack}[which]
  n, steps = 4, 2200
  eng = _engine(pack, n, auto_reset=True)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(8)
  restarts = 0
  for s in range(steps):
    acts = rng.integers(0, eng.num_actions, size=(n, eng.P), dtype=np.int32)
    eng.step(torch.from_numpy(acts).to(eng.device))
    for w, o in enumerate(oracles):
      if o.done:
        o.reset(); restarts += 1
      else:
        o.step(acts[w])
    if s % 50 == 0 or s == steps - 1:
      _compare_state(eng, oracles, f"step {s + 1}")
  _compare_rgb(eng, oracles, "end")
  assert restarts >= 1 and eng.counters()["episodes"] == n + restarts
  eng.close()


def test_a_step_on_a_finished_world_reports_nothing(coins_pack):
  """coins, world 835 of the deep soak's batch (tests/tools/deep_soak.py): the episode ends on the
  step that pays a coin; without auto-reset the steps that follow leave the world as it is and
  report no reward, no event, LAST and a zero discount (stepk::dispatch) — the oracle likewise
  (oracle_api.c frozen_step).  Its fifteen neighbours go on (or end on their own) beside it."""
  import torch
  from meltingpot_amd import engine as E
  first, n, steps = 832, 16, 520
  eng = _engine(coins_pack, n, auto_reset=False, world_offset=first)
  oracles = util.make_oracles(coins_pack, n, offset=first)
  eng.reset()
  for o in oracles:
    o.reset()
  worlds = np.arange(first, first + n)
  for s in range(steps):
    acts = util.hashed_actions(worlds, s, eng.P, num_actions=eng.num_actions)
    eng.step(torch.from_numpy(acts).to(eng.device))
    for w, o in enumerate(oracles):
      o.step(acts[w])
    if s >= 495:
      _compare_state(eng, oracles, f"step {s}")
      _compare_scalars(eng, oracles, f"step {s}")
      ev = eng.observe(E.OBS_EVENTS).cpu().numpy()
      for w, o in enumerate(oracles):
        got = sorted(tuple(int(v) for v in r[:3]) for r in ev[w, 1:1 + int(ev[w, 0, 0])])
        assert got == sorted(o.events()), (s, w, got, o.events())
    if s == 498:
      assert oracles[3].done and oracles[3].rewards().tolist() == [0.0, 1.0]
  st = eng.observe(E.OBS_STEP_TYPE).cpu().numpy()
  disc = eng.observe(E.OBS_DISCOUNT).cpu().numpy()
  assert st[3] == 2 and disc[3] == 0.0 and st[2] == 1 and disc[2] == 1.0
  eng.close()


def _packs_by_visible_planes():
  """Packs whose render planes that can show anything (DevTables::vis_layers: the planes the
  renderers read, by code unrolled for exactly their count — csrc/frame.hip `resolve`) number
  1 ... 12 (a pack whose avatars have no sprite is refused by mp_create): clean_up (7 of 9) with the sprites of whole layers taken away, territory__rooms
  (9 of 11) less one, collaborative_cooking__crowded (11 of 12) less one, as it is, and with
  a sprite for the one state of its logic layer."""
  from meltingpot_amd import engine as E, lower, pack as P
  def strip(name, drop, add=()):
    pb = E.load_pack(name)
    t = P.loads(pb)
    sprites = t["state_sprite"].copy()
    for i, l in enumerate(t["state_layer"]):
      if int(l) in drop:
        sprites[i] = -1
      if int(l) in add:
        sprites[i] = 1
    pb = util.patch_pack(pb, tables={"state_sprite": sprites})
    t = P.loads(pb)
    L = int(t["hdr"][lower.HDR_L])
    # (a sprite without a visible pixel — include/mp_pack.h MPK_SPRITE_EMPTY = 4 — shows nothing)
    vis = sorted({int(l) for l, sp in zip(t["state_layer"], t["state_sprite"])
                  if sp >= 0 and 0 <= l < L and not (int(t["sprite_flags"][sp]) & 4)})
    return name, pb, vis
  # clean_up: the avatars' plane stays (a pack whose avatars have no sprite is refused), the
  # others come back one at a time, from the bottom
  t = P.loads(E.load_pack("clean_up"))
  avatar_plane = int(t["state_layer"][int(t["avatar_alive_state"][0])])
  planes = sorted({int(l) for l, sp in zip(t["state_layer"], t["state_sprite"]) if sp >= 0 and l >= 0})
  others = [l for l in planes if l != avatar_plane]
  out = [strip("clean_up", set(others[k:])) for k in range(len(others) + 1)]
  def top_plane_without_avatars(name):
    t = P.loads(E.load_pack(name))
    ap = {int(t["state_layer"][int(a)]) for a in t["avatar_alive_state"]}
    return max(int(l) for l, sp in zip(t["state_layer"], t["state_sprite"]) if sp >= 0 and int(l) not in ap)
  out.append(strip("territory__rooms", {top_plane_without_avatars("territory__rooms")}))
  out.append(strip("territory__rooms", set()))
  out.append(strip("collaborative_cooking__crowded", {top_plane_without_avatars("collaborative_cooking__crowded")}))
  out.append(strip("collaborative_cooking__crowded", set()))
  out.append(strip("collaborative_cooking__crowded", set(), add={0}))
  return out


def test_every_count_of_visible_planes():
  """The renderers' resolve is one of twelve straight-line variants, picked by how many render
  planes can show anything; the committed packs reach 2, 4, 5, 7, 9 and 11.  Here every count from
  one to all twelve, both views in one launch and the draw-only launch, against the oracle on
  the same (patched) pack; `MpInfo.visible_layers` is the mask the pack implies."""
  packs = _packs_by_visible_planes()
  assert [len(v) for _, _, v in packs] == list(range(1, 13))
  from meltingpot_amd import engine as E
  for name, pb, vis in packs:
    eng = _engine(pb, 4)
    assert eng.info.visible_layers == sum(1 << l for l in vis), (name, vis, eng.info.visible_layers)
    eng.close()
    _run(pb, n=24, steps=24, seed=len(vis), rgb_every=6, fused="both")
    _run(pb, n=9, steps=12, seed=len(vis), rgb_every=4, fused=None)


@pytest.mark.parametrize("which,view,n,dev", [
    ("clean_up", "world", 150, {"pace": 2}),
    ("clean_up", "both", 70, {"pace": 7, "feeders": 3}),
    ("commons", "agents", 90, {"pace": 4, "static_pct": 50, "max_groups": 4}),
    ("territory", "agents", 80, {"pace": 3, "batch_worlds": 1, "ring_batches": 6, "max_groups": 3}),
])
def test_paced_renderers(clean_up_pack, commons_pack, territory_pack, which, view, n, dev):
  """FramePlan::pace — what a renderer wave sleeps between two passes, mp_tune's throttle for a
  view the memory side serves unevenly — changes when a pass is drawn, never what: forced
  (MpDevOptions.pace = 1 + units), alone and next to the other plan dimensions, against the
  oracle; the plan reports it."""
  from meltingpot_amd import engine as E
  pack = {"clean_up": clean_up_pack, "commons": commons_pack, "territory": territory_pack}[which]
  eng = _engine(pack, n, dev=dev)
  eng.bind(E.OBS_WORLD_RGB if view == "world" else E.OBS_RGB)
  assert eng.plan["pace"] == dev["pace"] - 1
  eng.close()
  _run(pack, n=n, steps=30, seed=n, rgb_every=6, fused=view, unfused=False, dev=dev)


@pytest.mark.parametrize("which,view,n,dev", [
    ("clean_up", "world", 150, {"batch_worlds": 1, "ring_batches": 8, "team": 1}),
    ("clean_up", "world", 333, {"batch_worlds": 1, "ring_batches": 8, "team": 1, "max_groups": 75}),  # teams of 10 and 9
    ("clean_up", "both", 70, {"batch_worlds": 1, "ring_batches": 6, "team": 1, "max_groups": 13, "pace": 3}),
    ("clean_up", "world", 5, {"batch_worlds": 1, "ring_batches": 4, "team": 1}),        # fewer workgroups than XCDs
    ("clean_up", "agents", 1, {"batch_worlds": 1, "ring_batches": 4, "team": 1}),
    ("commons", "agents", 90, {"batch_worlds": 1, "ring_batches": 6, "team": 1, "max_groups": 4}),
    ("territory", "agents", 80, {"batch_worlds": 1, "ring_batches": 6, "feeders": 3, "team": 1, "max_groups": 16}),
    ("clean_up", "world", 100, {"batch_worlds": 1, "ring_batches": 8, "team": 1, "static_pct": 50}),  # pooled: refused
    ("clean_up", "world", 100, {"batch_worlds": 2, "ring_batches": 4, "team": 1}),                    # B > 1: refused
])
def test_worlds_dealt_to_xcd_teams(clean_up_pack, commons_pack, territory_pack, which, view, n, dev):
  """FramePlan::team — with single-world batches the workgroups of an XCD share one contiguous
  range of worlds and deal it among themselves (member j of m takes worlds j, j + m, ...), so
  that each XCD writes one compact front; which worlds a workgroup owns changes, nothing else:
  ragged team ends, teams of unequal size, fewer workgroups than XCDs, one world, a pause on
  top, against the oracle; a plan with a pool or with batches of several worlds does not take
  the option."""
  from meltingpot_amd import engine as E
  pack = {"clean_up": clean_up_pack, "commons": commons_pack, "territory": territory_pack}[which]
  _run(pack, n=n, steps=24, seed=n, rgb_every=6, fused=view, unfused=False, dev=dev)


def _stock_plan(which, views):
  """frame.hip plan_frame's stock geometry for a stepping launch (B, NB, feeders)."""
  if views == "world":
    return 4, 2, 4
  return 3, 2, (3 if which == "territory" else 6)


TUNER_PLANS = [
    # (name, MpDevOptions, expectation on MpInfo.plan_*) — mp_tune's five candidates
    # (csrc/mp_engine.hip mp_tune), each FORCED: with MpConfig.dev the plan is the caller's
    # and mp_tune keeps it
    ("stock ring", lambda B, NB, F: {"static_pct": 100},
     lambda p, B, NB, F: p["batch_worlds"] == B and p["ring_batches"] == NB and
     p["pooled_batches"] == 0 and not p["sc1_stores"] and p["feeders"] == F),
    ("single-world ring", lambda B, NB, F: {"batch_worlds": 1, "ring_batches": B * NB},
     lambda p, B, NB, F: p["batch_worlds"] == 1 and p["ring_batches"] == B * NB and
     p["pooled_batches"] == 0),
    ("single-world ring, half pooled",
     lambda B, NB, F: {"batch_worlds": 1, "ring_batches": B * NB, "static_pct": 50},
     lambda p, B, NB, F: p["batch_worlds"] == 1 and p["ring_batches"] == B * NB and
     p["pooled_batches"] > 0),
    ("sc1 stores", lambda B, NB, F: {"store_sc1": 1},
     lambda p, B, NB, F: p["sc1_stores"] == 1 and p["batch_worlds"] == B),
    ("half the feeders", lambda B, NB, F: {"feeders": F // 2},
     lambda p, B, NB, F: p["feeders"] == F // 2 and p["batch_worlds"] == B),
    # round 6: the single-world ring dealt to XCD teams
    ("single-world ring, XCD teams", lambda B, NB, F: {"batch_worlds": 1, "ring_batches": B * NB, "team": 1},
     lambda p, B, NB, F: p["batch_worlds"] == 1 and p["ring_batches"] == B * NB and p["pooled_batches"] == 0
     and p["xcd_teams"] == 1),
    # round 6: the feeders keep priority 1 after their first world
    ("stock ring, feeders at priority 1", lambda B, NB, F: {"late_feeder_prio": 2},
     lambda p, B, NB, F: p["late_feeder_priority"] == 1 and p["batch_worlds"] == B and p["feeders"] == F),
    # round 6: a renderer wave sleeps three units between two passes (MpDevOptions.pace = 1 + units)
    ("stock ring, paced", lambda B, NB, F: {"pace": 4},
     lambda p, B, NB, F: p["pace"] == 3 and p["batch_worlds"] == B and p["feeders"] == F),
]


@pytest.mark.parametrize("which,n,views", [
    ("clean_up", 4096, "world"),     # BASELINE.json configs[1]
    ("commons", 4096, "agents"),     # configs[2]
    ("territory", 8192, "agents"),   # configs[3]
    ("clean_up", 4096, "both"),      # bench.py's substrate_api
])
def test_tuner_plans_at_full_size(clean_up_pack, commons_pack, territory_pack, which, n, views):
  """Which launch plan a full-size run exercises must not be decided by a timer: every plan
  `mp_tune` can keep — the stock ring; the same LDS cut into single-world batches; that with
  half of every workgroup's share pooled behind the claim counter; sc1 pixel stores; half the
  feeders; a pause between the renderers' passes — is FORCED here (MpDevOptions) at BASELINE.json's batch sizes in the launch form
  bench.py times (the views bound before the first step), `MpInfo.plan_*` is checked to BE
  the forced plan, and after 64 steps 512 worlds (8 blocks of 64 across the batch: first,
  last, workgroup boundaries) are replayed by the oracle: state, rewards, events and the
  bound views bit-exact."""
  import torch
  from meltingpot_amd import engine as E
  pack = {"clean_up": clean_up_pack, "commons": commons_pack, "territory": territory_pack}[which]
  B, NB, F = _stock_plan(which, views)
  steps = 64
  blocks = [int(b) for b in np.linspace(0, n - 64, 8)]
  blocks[3] = (n // 2) - 32            # straddles the middle workgroups' boundary
  gen = torch.Generator(device="cuda")
  gen.manual_seed(4)
  acts = None
  want = {}                            # block -> oracle results (the same for every plan)
  ran = 0
  for name, make_dev, holds in TUNER_PLANS:
    if name == "sc1 stores" and views == "world":
      continue                         # (mp_tune offers it to the per-agent views only)
    if name == "half the feeders" and (views == "world" or F < 4):
      continue
    eng = _engine(pack, n, dev=make_dev(B, NB, F), placements=0)
    if acts is None:
      acts = torch.randint(0, eng.num_actions, (steps, n, eng.P), generator=gen,
                           device=eng.device, dtype=torch.int32)
      host_acts = acts.cpu().numpy()
    bound = {}
    if views in ("agents", "both"):
      bound[E.OBS_RGB] = eng.bind(E.OBS_RGB)
    if views in ("world", "both"):
      bound[E.OBS_WORLD_RGB] = eng.bind(E.OBS_WORLD_RGB)
    assert eng.fused
    eng.tune()                         # (explicit plans: a no-op that must stay one)
    assert holds(eng.plan, B, NB, F), (name, eng.plan)
    eng.reset()
    for s in range(steps):
      eng.step(acts[s])
    assert holds(eng.plan, B, NB, F), (name, eng.plan)
    grid, avat, glob = eng.dump()
    rew = eng.observe(E.OBS_REWARD).cpu().numpy()
    ev = eng.observe(E.OBS_EVENTS).cpu().numpy()
    for b in blocks:
      if b not in want:
        want[b] = list(util.replay_parallel(pack, host_acts[:, b:b + 64], looks=(steps,),
                                            sample=range(b, b + 64), world_view="both", offset=b))
      for w, og, oa, ogl, orew, oev, looks in want[b]:
        assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa), (name, w)
        assert np.array_equal(glob[w], ogl) and np.array_equal(rew[w], orew), (name, w)
        got = sorted(tuple(int(v) for v in r[:3]) for r in ev[w, 1:1 + int(ev[w, 0, 0])])
        assert got == oev, (name, w)
      world_px = np.stack([l[steps][0] for *_, l in want[b]])
      agent_px = np.stack([l[steps][1] for *_, l in want[b]])
      if E.OBS_WORLD_RGB in bound:
        assert np.array_equal(bound[E.OBS_WORLD_RGB][b:b + 64].cpu().numpy(), world_px), (name, b)
      if E.OBS_RGB in bound:
        assert np.array_equal(bound[E.OBS_RGB][b:b + 64].cpu().numpy(), agent_px), (name, b)
    assert not eng.fault_words()[:6].any()
    eng.close()
    ran += 1
  assert ran >= (3 if views == "world" else 4)
