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
  with pytest.raises(ValueError):
    substrate.build("bach_or_stravinsky_in_the_matrix__repeated", roles=("default", "default"))
