"""GPU parity of the *_in_the_matrix level (csrc/step_matrix.h) through the C ABI:
state, rewards, READY_TO_SHOOT, INVENTORY, INTERACTION_INVENTORIES, events and both
RGB views bit-exact against the CPU oracle (oracle/the_matrix.c), episodes
restarting as they end."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

# interaction-heavy: NOOP FORWARD BACKWARD STEP_LEFT STEP_RIGHT TURN_LEFT TURN_RIGHT INTERACT
WEIGHTS = [1, 6, 1, 1, 1, 2, 2, 5]


def _run(name, n, steps, seed, weights=WEIGHTS, rgb_every=25, bind=("world",), players=0,
         max_frames=None, stats=None, variant=None, **engine_kw):
  import torch
  from meltingpot_amd import engine as E
  assert torch.cuda.is_available(), "gpu tests need a GPU"
  pack = E.load_pack(name)
  if variant:
    pack = util.matrix_variant(pack, **variant)
  if max_frames:
    pack = util.patch_pack(pack, MAXFRAMES=max_frames)
  eng = E.Engine(pack, n, num_players=players, auto_reset=True, **engine_kw)
  bound = {}
  if "world" in bind:
    bound[E.OBS_WORLD_RGB] = eng.bind(E.OBS_WORLD_RGB)
  if "agents" in bind:
    bound[E.OBS_RGB] = eng.bind(E.OBS_RGB)
  oracles = util.make_oracles(pack, n, num_players=players)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(seed)
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions, weights)
  dacts = torch.from_numpy(acts).to(eng.device)
  interactions = 0

  def compare(tag, rgb):
    nonlocal interactions
    grid, avat, glob = eng.dump()
    rew = eng.observe(E.OBS_REWARD).cpu().numpy()
    rdy = eng.observe(E.OBS_READY_TO_SHOOT).cpu().numpy()
    inv = eng.observe(E.OBS_INVENTORY).cpu().numpy()
    inter = eng.observe(E.OBS_INTERACTION_INVENTORIES).cpu().numpy()
    col = eng.observe(E.OBS_COLLECTIVE_REWARD).cpu().numpy()
    for w, o in enumerate(oracles):
      og, oa, ogl = o.dump()
      assert np.array_equal(glob[w], ogl), (tag, w, glob[w], ogl)
      if not np.array_equal(avat[w], oa):
        raise AssertionError(f"{tag}: world {w} avatars differ:\n{avat[w]}\n{oa}")
      if not np.array_equal(grid[w], og):
        bad = np.argwhere(grid[w] != og)
        raise AssertionError(f"{tag}: world {w} grid differs at (layer, y, x) {bad[:6].tolist()}: "
                             f"gpu {grid[w][tuple(bad[0])]} oracle {og[tuple(bad[0])]}")
      assert np.array_equal(rew[w], o.rewards()), (tag, w, rew[w], o.rewards())
      assert np.array_equal(rdy[w], o.ready_to_shoot()), (tag, w)
      if stats is not None:   # markers away from their (live) avatars, or off the grid
        v, alive = oa[:, 7], oa[:, 3] == 1
        on = (v & 1) == 1
        stats["detached"] = stats.get("detached", 0) + int(np.sum(
            alive & on & ((((v >> 1) & 255) != oa[:, 0]) | (((v >> 9) & 255) != oa[:, 1]))))
        stats["off_grid"] = stats.get("off_grid", 0) + int(np.sum(alive & ~on))
      oinv, ointer = o.inventories()
      assert np.array_equal(inv[w], oinv), (tag, w, inv[w], oinv)
      assert np.array_equal(inter[w], ointer), (tag, w, inter[w], ointer)
      assert col[w] == o.rewards().sum(), (tag, w)
      ev = eng.events(w)
      oev = o.events()
      interactions += sum(e[0] == 11 for e in oev)
      assert sorted((E_name(t), a, b) for t, a, b in oev) == sorted(
          (nm, *_payload(pl)) for nm, pl in ev), (tag, w, ev, oev)
      # the 'interaction' event's whole payload (the_matrix/components.lua:789-797):
      # exact doubles, the oracle's
      orew = o.interaction_rewards()
      for nm, pl in ev:
        if nm != "interaction":
          continue
        r, c = pl["row_player_idx"] - 1, pl["col_player_idx"] - 1
        assert list(pl) == ["row_player_idx", "col_player_idx", "row_reward", "col_reward",
                            "row_inventory", "col_inventory"]
        assert (pl["row_reward"], pl["col_reward"]) == (orew[r, 0], orew[r, 1]) == (
            orew[c, 0], orew[c, 1]), (tag, w, pl, orew)
        assert np.array_equal(pl["row_inventory"], ointer[r, 0]), (tag, w, pl)
        assert np.array_equal(pl["col_inventory"], ointer[c, 0]), (tag, w, pl)
        assert np.array_equal(pl["col_inventory"], ointer[r, 1]), (tag, w, pl)
    if rgb:
      wrgb = (bound.get(E.OBS_WORLD_RGB) if E.OBS_WORLD_RGB in bound
              else eng.observe(E.OBS_WORLD_RGB)).cpu().numpy()
      argb = (bound.get(E.OBS_RGB) if E.OBS_RGB in bound else eng.observe(E.OBS_RGB)).cpu().numpy()
      for w, o in enumerate(oracles):
        ow = o.render_world()
        if not np.array_equal(wrgb[w], ow):
          bad = np.argwhere(wrgb[w] != ow)
          raise AssertionError(f"{tag}: WORLD.RGB world {w} differs at {bad[:4].tolist()}")
        for p in range(o.P):
          oa = o.render_agent(p)
          if not np.array_equal(argb[w, p], oa):
            bad = np.argwhere(argb[w, p] != oa)
            raise AssertionError(f"{tag}: RGB world {w} player {p} differs at {bad[:4].tolist()}")

  from meltingpot_amd.engine import EVENT_TYPES

  def E_name(t):
    return EVENT_TYPES[t][0]

  def _payload(pl):
    vals = list(pl.values()) + [0, 0]
    return vals[0], vals[1]

  compare("reset", True)
  for s in range(steps):
    eng.step(dacts[s])
    for w, o in enumerate(oracles):
      if o.done:
        o.reset()
      else:
        o.step(acts[s, w])
    compare(f"step {s + 1}", (s + 1) % rgb_every == 0 or s == steps - 1)
  eng.close()
  return interactions


def test_prisoners_dilemma_repeated_rollout():
  assert _run("prisoners_dilemma_in_the_matrix__repeated", n=16, steps=400, seed=1) > 0


@pytest.mark.parametrize("unfused", [False, None])
@pytest.mark.parametrize("bind", [("world",), ("agents",), ("world", "agents"), ()])
def test_launch_forms_agree(bind, unfused):
  """The bound view is drawn by the launch that steps the worlds (k_frame), the
  other one from the stepped records; no view bound: the stand-alone step kernel.
  unfused=False forces the fused launch (the engine's own choice for the small
  views of the two-player games is two launches)."""
  _run("prisoners_dilemma_in_the_matrix__repeated", n=6, steps=120, seed=2, bind=bind,
       unfused=unfused)


def test_unfused_launches_agree():
  _run("chicken_in_the_matrix__repeated", n=6, steps=150, seed=3, unfused=True)


@pytest.mark.parametrize("name", [
    "prisoners_dilemma_in_the_matrix__arena", "stag_hunt_in_the_matrix__arena",
    "running_with_scissors_in_the_matrix__arena", "bach_or_stravinsky_in_the_matrix__arena",
    "pure_coordination_in_the_matrix__arena", "chicken_in_the_matrix__arena",
    "rationalizable_coordination_in_the_matrix__arena"])
def test_arena_rollouts(name):
  """8 players: simultaneous interactions, detached markers, crowded respawns."""
  assert _run(name, n=12, steps=500, seed=4, rgb_every=50) > 0


@pytest.mark.parametrize("name", [
    "chicken_in_the_matrix__repeated", "stag_hunt_in_the_matrix__repeated",
    "pure_coordination_in_the_matrix__repeated",
    "rationalizable_coordination_in_the_matrix__repeated",
    "bach_or_stravinsky_in_the_matrix__repeated",
    "running_with_scissors_in_the_matrix__repeated",
    "running_with_scissors_in_the_matrix__one_shot"])
def test_two_player_rollouts(name):
  _run(name, n=24, steps=500, seed=5, rgb_every=100, unfused=False)


def test_long_rollout_through_episode_ends():
  """Episodes end stochastically after frame 1000 (and at maxEpisodeLengthFrames):
  the worlds restart with the next episode's draws."""
  assert _run("prisoners_dilemma_in_the_matrix__arena", n=8, steps=2600, seed=6,
              rgb_every=400) > 0


@pytest.mark.parametrize("name", ["prisoners_dilemma_in_the_matrix__arena",
                                  "stag_hunt_in_the_matrix__arena"])
def test_detached_readiness_markers(name):
  """A respawning avatar's marker is first put where it was left; if another
  marker stands there the placement fails, is retried by the priority-2 updater
  and the marker can end up away from its avatar, following it in parallel (A14,
  A18).  Rare in random play: this rollout is long and crowded enough that it
  happens (asserted), and the GPU has to reproduce every such frame."""
  stats = {}
  _run(name, n=32, steps=1500, seed=9, weights=[1, 8, 1, 1, 1, 2, 2, 6], rgb_every=250,
       stats=stats)
  assert stats["detached"] > 0 and stats["off_grid"] > 0, stats


def test_full_size_batch():
  """The batch sizes the bench runs this level at: EVERY one of 4096 arena worlds
  replayed by the oracle for 48 steps (state, rewards, inventories, events), the
  per-agent view on the last step."""
  assert _run("prisoners_dilemma_in_the_matrix__arena", n=4096, steps=48, seed=10,
              rgb_every=48, bind=("agents",)) > 0


def test_small_views_fused_in_batches_of_eight():
  """The two-player games draw 2 x 40 x 40 pixels a world; since round 3 their step
  is fused with the drawing too, in batches of 8 worlds with 8 feeders (plan_frame).
  5000 worlds = 20 worlds a workgroup: three batches through the two-buffer ring,
  the last one partial; every world against the oracle, episodes restarting."""
  assert _run("prisoners_dilemma_in_the_matrix__repeated", n=5000, steps=40, seed=21,
              rgb_every=20, bind=("agents",), max_frames=25) >= 0


@pytest.mark.parametrize("name,variant", [
    # Taste pays for gathering, the zapped player's InteractionTaste prices both
    # rewards, a multiplier, a penalty for zapping unready players
    ("prisoners_dilemma_in_the_matrix__arena",
     dict(taste=[(1, 0.5, 0.125), (2, 0.25, 0.0), (-1, 1.0, 0.0)],
          itaste=[(1, True, 1.5), (2, False, 0.75), (-1, False, 0.0)],
          multiplier=0.5, unready=-0.25)),
    ("running_with_scissors_in_the_matrix__repeated",
     dict(taste=[(3, 1.0, 0.0), (1, 0.5, 0.25)], itaste=[(3, True, 2.0), (2, False, 1.0)],
          unready=-1.0, regen_rate=0.2)),
    # inventories start at 0, ties broken by a draw, rewards under a floor withheld
    ("stag_hunt_in_the_matrix__arena", dict(zero_inventory=True, random_tie=True, floor=2.5)),
    ("pure_coordination_in_the_matrix__repeated",
     dict(zero_inventory=True, random_tie=True, regen_rate=0.3)),
])
def test_rule_constants_the_stock_configs_leave_at_their_defaults(name, variant):
  """util.matrix_variant: the branches of Taste, InteractionTaste, rewardFloor,
  rewardMultiplier, rewardFromZappingUnreadyPlayer, zeroInitialInventory and
  randomTieBreaking that no stock pack takes."""
  assert _run(name, n=16, steps=600, seed=12, rgb_every=200, variant=variant) > 0


def test_short_episodes_restart_often():
  _run("running_with_scissors_in_the_matrix__arena", n=8, steps=300, seed=7, max_frames=40,
       rgb_every=60)


def test_fewer_players_than_the_pack_holds():
  _run("prisoners_dilemma_in_the_matrix__arena", n=8, steps=300, seed=8, players=5)


def test_substrate_api_shapes():
  import torch
  from meltingpot_amd import substrate
  env = substrate.build("prisoners_dilemma_in_the_matrix__repeated", roles=("default",) * 2,
                        num_worlds=4)
  ts = env.reset()
  assert ts.observation["RGB"].shape == (4, 2, 40, 40, 3)
  assert ts.observation["INVENTORY"].shape == (4, 2, 2)
  assert ts.observation["INTERACTION_INVENTORIES"].shape == (4, 2, 2, 2)
  assert ts.observation["WORLD.RGB"].shape == (4, 120, 184, 3)
  ts = env.step(torch.zeros((4, 2), dtype=torch.int32, device=ts.observation["RGB"].device))
  assert torch.all(ts.observation["INVENTORY"] == 1.0)
  assert torch.all(ts.observation["INTERACTION_INVENTORIES"] == -1.0)
  env.close()
  # one world, numpy leaves, the reference's per-player list
  env = substrate.build("bach_or_stravinsky_in_the_matrix__repeated",
                        roles=("bach_fan", "stravinsky_fan"))
  ts = env.reset()
  assert ts.observation[0]["INVENTORY"].shape == (2,) and ts.observation[1]["RGB"].shape == (40, 40, 3)
  env.close()
  # "default" is a valid role there, resolved per player index
  # (bach_or_stravinsky_in_the_matrix__repeated.py:478-484)
  with substrate.build("bach_or_stravinsky_in_the_matrix__repeated",
                       roles=("default", "default")) as env:
    assert env.reset().observation[0]["RGB"].shape == (40, 40, 3)
  with pytest.raises(ValueError):
    substrate.build("bach_or_stravinsky_in_the_matrix__repeated", roles=("bach_fan", "wagner_fan"))


@pytest.mark.parametrize("name,roles", [
    ("bach_or_stravinsky_in_the_matrix__repeated", ("default", "default")),
    ("bach_or_stravinsky_in_the_matrix__repeated", ("stravinsky_fan", "bach_fan")),
    ("bach_or_stravinsky_in_the_matrix__repeated", ("bach_fan", "bach_fan")),
    ("bach_or_stravinsky_in_the_matrix__arena",
     ("stravinsky_fan", "default", "bach_fan", "bach_fan", "default", "stravinsky_fan",
      "default", "bach_fan")),
    ("bach_or_stravinsky_in_the_matrix__arena", ("stravinsky_fan", "bach_fan", "default")),
])
def test_role_assignments(name, roles):
  """MpConfig.roles: the engine created for an assignment of roles
  (create_avatar_objects(roles), bach_or_stravinsky_in_the_matrix__repeated.py:
  473-497: row / column player and avatar colour per role; "default" = by player
  index) steps and renders like the oracle on the pack that assignment lowers to
  (lower.apply_roles; tests/test_oracle_matrix_cpu.py checks that pack against
  the reference's own build(roles))."""
  import torch
  from meltingpot_amd import engine as E, lower, pack as pack_lib
  base = E.load_pack(name)
  names = E.pack_role_names(base)
  assert names == ("bach_fan", "default", "stravinsky_fan")
  ids = [names.index(r) for r in roles]
  want = pack_lib.dumps(lower.apply_roles(pack_lib.loads(base), ids))
  n, steps = 6, 400
  eng = E.Engine(base, n, roles=ids, auto_reset=True)
  assert eng.P == len(roles)
  wrgb = eng.bind(E.OBS_WORLD_RGB)
  oracles = util.make_oracles(want, n, num_players=len(roles))
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(len(roles))
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions, WEIGHTS)
  interactions = 0
  for s in range(steps):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      if o.done:
        o.reset()
      else:
        o.step(acts[s, w])
    if s % 20 == 19:
      grid, avat, glob = eng.dump()
      rew = eng.observe(E.OBS_REWARD).cpu().numpy()
      rgb = eng.observe(E.OBS_RGB).cpu().numpy()
      inter = eng.observe(E.OBS_INTERACTION_INVENTORIES).cpu().numpy()
      for w, o in enumerate(oracles):
        og, oa, ogl = o.dump()
        assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa), (s, w)
        assert np.array_equal(glob[w], ogl) and np.array_equal(rew[w], o.rewards()), (s, w)
        assert np.array_equal(wrgb[w].cpu().numpy(), o.render_world()), (s, w)
        for p in range(o.P):
          assert np.array_equal(rgb[w, p], o.render_agent(p)), (s, w, p)
        assert np.array_equal(inter[w], o.inventories()[1]), (s, w)
    interactions = eng.counters()["aux0"]
  same_role = len(set(r if r != "default" else ("bach_fan", "stravinsky_fan")[i % 2]
                      for i, r in enumerate(roles))) == 1
  # two row players (or two column players) never resolve an interaction (:773-785)
  if same_role:
    assert interactions == 0, (interactions, roles)
  elif len(roles) >= 8:    # (two random players on the small map rarely meet in 400 steps)
    assert interactions > 0, roles
  eng.close()
  # the same through the drop-in API, roles by name
  from meltingpot_amd import substrate
  with substrate.build(name, roles=roles, num_worlds=2, env_seed=3) as env:
    assert env.num_players == len(roles)
    env.reset()


@pytest.mark.parametrize("name", ["prisoners_dilemma_in_the_matrix__arena",
                                  "running_with_scissors_in_the_matrix__repeated",
                                  "bach_or_stravinsky_in_the_matrix__arena"])
def test_debug_cumulants(name):
  """MP_OBS_MATRIX_CUMULANTS = the_matrix.get_cumulant_metric_configs
  (the_matrix.py:22-60): INTERACTED_THIS_STEP, COLLECTED_RESOURCE_k,
  DESTROYED_RESOURCE_k, ARGMAX_INTERACTION_INVENTORY_WAS_k per player, every step,
  bound and through debug_observations, against the oracle's."""
  import torch
  from meltingpot_amd import engine as E
  pack = E.load_pack(name)
  n, steps = 8, 500
  eng = E.Engine(pack, n, auto_reset=True)
  other = E.Engine(pack, n, auto_reset=True, debug_observations=True)
  with pytest.raises(E.EngineError):
    eng.observe(E.OBS_MATRIX_CUMULANTS)       # a debug observation: not produced unless asked for
  cum = eng.bind(E.OBS_MATRIX_CUMULANTS)
  R = eng.info.num_resources
  assert cum.shape == (n, eng.P, 1 + 3 * R)
  assert E.matrix_cumulant_names(R)[:4] == ["INTERACTED_THIS_STEP", "COLLECTED_RESOURCE_1",
                                            "DESTROYED_RESOURCE_1",
                                            "ARGMAX_INTERACTION_INVENTORY_WAS_1"]
  oracles = util.make_oracles(pack, n)
  eng.reset(); other.reset()
  for o in oracles:
    o.reset()
  assert not cum.any()
  rng = np.random.default_rng(12)
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions, WEIGHTS)
  seen = np.zeros(1 + 3 * R)
  for s in range(steps):
    a = torch.from_numpy(acts[s]).to(eng.device)
    eng.step(a); other.step(a)
    got = cum.cpu().numpy()
    assert np.array_equal(got, other.observe(E.OBS_MATRIX_CUMULANTS).cpu().numpy()), s
    for w, o in enumerate(oracles):
      if o.done:
        o.reset()
        assert not got[w].any(), (s, w)
        continue
      o.step(acts[s, w])
      assert np.array_equal(got[w], o.matrix_cumulants()), (s, w, got[w], o.matrix_cumulants())
    seen += got.sum(axis=(0, 1))
  assert seen[0] > 0 and seen[1] > 0 and seen[3] > 0 and seen[2::3].sum() > 0, seen
  eng.close(); other.close()


def test_colour_intervals_must_hold_every_reward():
  """TheMatrix:getColorInterval asserts (components.lua:282-290).  A pack whose
  intervals leave a gap inside the payable range is refused by mp_create; a
  reward on the very end of the range (the stock intervals are half-open there)
  is reported once by the next synchronising call, and the engine goes on."""
  import torch
  from meltingpot_amd import engine as E, pack as pack_lib
  name = "prisoners_dilemma_in_the_matrix__repeated"
  t = pack_lib.loads(E.load_pack(name))
  R, NI = int(t["mx_i32"][0]), int(t["mx_i32"][19])
  f = t["mx_f64"].copy()
  iv = f[5 + 2 * R * R:5 + 2 * R * R + 2 * NI].reshape(NI, 2)
  iv[1] = (1.5, 2.0)                                 # nothing holds [1.0, 1.5)
  with pytest.raises(E.EngineError, match="resultIndicatorColorIntervals"):
    E.Engine(util.patch_pack(E.load_pack(name), tables={"mx_f64": f}), 2)
  # every payoff 5.0: just outside the last interval [4, 5)
  f = t["mx_f64"].copy()
  f[5:5 + 2 * R * R] = 5.0
  eng = E.Engine(util.patch_pack(E.load_pack(name), tables={"mx_f64": f}), 16, auto_reset=True)
  eng.reset()
  rng = np.random.default_rng(0)
  acts = util.random_actions(rng, 400, 16, eng.P, eng.num_actions, WEIGHTS)
  for s in range(400):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
  with pytest.raises(E.EngineError, match="outside every resultIndicatorColorInterval"):
    eng.sync()
  eng.sync()                                 # reported once; the engine goes on
  assert eng.counters()["aux0"] > 0          # interactions did happen
  eng.close()
