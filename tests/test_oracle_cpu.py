"""CPU checks of the oracle itself (test infrastructure) and of the committed
artefacts: the pack is what the reference config lowers to, the golden fixture
is reproducible, and the rule restatement behaves as the reference Lua says."""
import hashlib
import json
import os

import numpy as np
import pytest

import util
from meltingpot_amd import engine, lower, pack, refshim
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "clean_up_1000_steps.json")


@pytest.mark.skipif(not os.path.isdir(refshim.DEFAULT_REFERENCE_ROOT),
                    reason="reference tree not present (GPU box)")
def test_committed_pack_is_what_the_reference_config_lowers_to(clean_up_pack):
  # tools/make_packs.py: lowered for the config's 15 avatar colours, 7 play by default
  settings, mod, _ = refshim.build_settings("clean_up", ("default",) * 15)
  blob = pack.dumps(lower.lower("clean_up", settings, mod.ACTION_SET, default_players=7))
  assert blob == clean_up_pack, "run tools/make_packs.py"


def test_pack_round_trip(clean_up_pack):
  t = pack.loads(clean_up_pack)
  assert pack.dumps(t) == clean_up_pack
  hdr = t["hdr"]
  # clean_up.py:55-77 (21x30 map), :855 spriteSize 8, 7 players, 9 actions
  assert (hdr[lower.HDR_H], hdr[lower.HDR_W]) == (21, 30)
  assert hdr[lower.HDR_P] == 15 and hdr[lower.HDR_DEFAULT_P] == 7 and hdr[lower.HDR_NACT] == 9
  assert hdr[lower.HDR_L] == 9 and hdr[lower.HDR_SPRITE] == 8
  # 122 potential apples, 147 dirt containers, 167 water
  assert len(t["apple_cells"]) == 122 and len(t["dirt_cells"]) == 147
  assert len(t["water_cells"]) == 167 and len(t["spawn_cells"]) == 19


def test_prob_threshold_is_the_exact_double_compare():
  # u < p for u = r * 2^-53  <=>  r < ceil(p * 2^53)
  rng = np.random.default_rng(0)
  for p in [0.05, 0.5, 0.2, 1e-9, 0.999999, 0.05 * (1 - 79 / 147)]:
    thr = lower.prob_threshold(p)
    for r in list(rng.integers(0, 1 << 53, 200)) + [thr - 1, thr, thr + 1, 0]:
      r = int(min(max(r, 0), (1 << 53) - 1))
      assert ((r * 2.0**-53) < p) == (r < thr)
  assert lower.prob_threshold(0.0) == 0 and lower.prob_threshold(-1.0) == 0
  assert lower.prob_threshold(1.5) == 1 << 53


def _rollout(pack_bytes, seed, steps, nact=9, with_events=False):
  o = oracle.Oracle(pack_bytes, util.world_seed(0))
  o.reset()
  rng = np.random.default_rng(seed)
  acts = rng.integers(0, nact, size=(steps, o.P), dtype=np.int32)
  h = hashlib.sha256()
  rewards = np.zeros(o.P)
  for s in range(steps):
    o.step(acts[s])
    grid, avat, glob = o.dump()
    h.update(grid.tobytes()); h.update(avat.tobytes()); h.update(glob.tobytes())
    rewards += o.rewards()
    if with_events:
      h.update(repr(o.events()).encode())
    if (s + 1) % 100 == 0:
      h.update(o.render_world().tobytes())
      for p in range(o.P):
        h.update(o.render_agent(p).tobytes())
  return h.hexdigest(), rewards, o


def test_golden_1000_step_fixture(clean_up_pack):
  """BASELINE.json configs[0]: clean_up, 7 players, 1 world, 1000 fixed-seed
  steps on the CPU path.  The fixture (tests/golden, made by
  tests/tools/make_golden.py) pins the oracle across refactors; the GPU engine is
  pinned to the oracle by tests/test_gpu_parity.py."""
  want = json.load(open(GOLDEN))
  got, rewards, o = _rollout(clean_up_pack, want["action_seed"], want["steps"])
  assert got == want["sha256"]
  fert = util.fertile_clean_up(clean_up_pack)
  got2, rewards2, _ = _rollout(fert, want["action_seed"], want["steps"])
  assert got2 == want["sha256_fertile"]
  assert rewards2.sum() == want["fertile_reward_sum"] > 0


@pytest.mark.parametrize("name,nact", [("commons_harvest__open", 8), ("territory__rooms", 9),
                                       ("coop_mining", 8), ("gift_refinements", 9),
                                       ("collaborative_cooking__cramped", 8),
                                       ("collaborative_cooking__crowded", 8),
                                       ("externality_mushrooms__dense", 8)])
def test_golden_fixtures_of_the_other_levels(name, nact):
  """Same recipe for BASELINE.json's other two levels and for coop_mining and gift_refinements (events
  included in the hash): the fixtures freeze the restated commons_harvest /
  territory / coop_mining / gift_refinements rules."""
  from meltingpot_amd import engine
  want = json.load(open(os.path.join(os.path.dirname(GOLDEN), f"{name}_1000_steps.json")))
  got, rewards, _ = _rollout(engine.load_pack(name), want["action_seed"], want["steps"],
                             nact=nact, with_events=True)
  assert got == want["sha256"] and rewards.sum() == want["reward_sum"]


def test_oracle_is_deterministic_and_seed_sensitive(clean_up_pack):
  # builder_test.py:47-106: same seed => same WORLD.RGB; other seed differs
  a = oracle.Oracle(clean_up_pack, 5); a.reset()
  b = oracle.Oracle(clean_up_pack, 5); b.reset()
  c = oracle.Oracle(clean_up_pack, 6); c.reset()
  assert np.array_equal(a.render_world(), b.render_world())
  assert not np.array_equal(a.render_world(), c.render_world())
  # The reference rebuilds with seed + 1 on reset (builder.py:177-181), so there
  # episode 1 of seed 5 IS episode 0 of seed 6 — with one seed per world of a
  # batch, neighbours would replay each other's randomness one episode later.
  # Here (world seed, episode) keys the generator: every pair is its own stream.
  first = a.render_world().copy()
  a.reset()
  assert not np.array_equal(a.render_world(), c.render_world())
  assert not np.array_equal(a.render_world(), first)
  b.reset()
  assert np.array_equal(a.render_world(), b.render_world())


def test_observation_shapes_match_the_reference_specs(clean_up_pack):
  o = oracle.Oracle(clean_up_pack, 1); o.reset()
  assert o.render_agent(0).shape == (88, 88, 3)      # specs.py:39
  assert o.render_world().shape == (168, 240, 3)     # clean_up.py:831
  assert o.ready_to_shoot().tolist() == [1.0] * 7    # avatar_library.lua:737-744


def test_zap_removes_and_respawns_after_50_frames(clean_up_pack):
  """Zapper (avatar_library.lua:613-681): find a world/step where a zap lands,
  check the victim leaves the grid, READY_TO_SHOOT follows the cooldown, and
  the victim is back 50 frames later (clean_up.py:707-716)."""
  o = oracle.Oracle(clean_up_pack, util.world_seed(3)); o.reset()
  rng = np.random.default_rng(5)
  zapped_at = {}
  delays = []
  for s in range(400):
    acts = rng.choice(9, size=7, p=np.array([1, 3, 1, 1, 1, 2, 2, 6, 1]) / 18.0).astype(np.int32)
    before = o.dump()[1][:, 3].copy()
    rdy_before = o.ready_to_shoot()
    o.step(acts)
    after = o.dump()[1]
    for p in range(7):
      if before[p] == 1 and after[p, 3] == 0:
        zapped_at[p] = s
      if before[p] == 0 and after[p, 3] == 1:
        # exactly framesTillRespawn unless the drawn spawn point was occupied,
        # in which case the updater retries on the next frame (assumption A5)
        assert s - zapped_at[p] >= 50, (p, s, zapped_at[p])
        delays.append(s - zapped_at[p])
      if acts[p] == 7 and before[p] == 1 and rdy_before[p] == 1.0:
        assert o.ready_to_shoot()[p] == 0.0 or after[p, 3] == 0
  assert zapped_at, "no zap landed in 400 steps"
  assert delays and min(delays) == 50 and max(delays) <= 53


def test_cleaning_reduces_dirt_and_counts_others(clean_up_pack):
  o = oracle.Oracle(clean_up_pack, util.world_seed(1)); o.reset()
  rng = np.random.default_rng(0)
  dirt0 = int(o.dump()[2][3])
  assert dirt0 == 79  # 'F' cells start dirty
  seen_metric = False
  for s in range(300):
    o.step(rng.choice(9, size=7, p=np.array([1, 2, 1, 1, 1, 1, 1, 0, 10]) / 18.0).astype(np.int32))
    m = o.num_others_cleaned()
    seen_metric |= bool(m.max() > 0)
    assert m.max() <= 6
  assert seen_metric


# ---------------------------------------------------------------- commons_harvest__open


@pytest.mark.skipif(not os.path.isdir(refshim.DEFAULT_REFERENCE_ROOT),
                    reason="reference tree not present (GPU box)")
def test_committed_commons_pack_is_what_the_reference_config_lowers_to(commons_pack):
  settings, mod, _ = refshim.build_settings("commons_harvest__open", ("default",) * 16)
  blob = pack.dumps(lower.lower("commons_harvest__open", settings, mod.ACTION_SET))
  assert blob == commons_pack, "run tools/make_packs.py"


def test_commons_pack_constants(commons_pack):
  t = pack.loads(commons_pack)
  hdr = t["hdr"]
  # commons_harvest__open.py:60-79 (18x24 map), 16 players, 8 actions
  assert (hdr[lower.HDR_H], hdr[lower.HDR_W]) == (18, 24)
  assert hdr[lower.HDR_P] == 16 and hdr[lower.HDR_NACT] == 8 and hdr[lower.HDR_L] == 8
  assert len(t["apple_cells"]) == 64
  assert len(t["spawn_cells"]) == 60                      # 'P' cells
  assert t["init_spawn_ptr"].tolist() == [0, 2, 62]       # 2 'Q' cells first
  assert t["avatar_init_group"].tolist() == [0, 0] + [1] * 14  # :520-528
  assert t["ch_i32"][0] == 14                             # floor(pi*4+1)+1 wait states
  assert len(t["disc_offsets"]) // 2 == 12                # L2 disc of radius 2
  # REGROWTH_PROBABILITIES = [0.0, 0.0025, 0.005, 0.025] (:57-58)
  thr = t["ch_thr"]
  assert thr[0] == 0 and thr[1] == lower.prob_threshold(0.0025)
  assert thr[3] == thr[13] == lower.prob_threshold(0.025)
  assert thr[14] == lower.prob_threshold(0.15)


def test_commons_density_regrow_invariants(commons_pack):
  """DensityRegrow (components.lua:161-240): a waiting apple's state index is the
  number of live apples within the radius-2 disc as of the previous frame; with
  no live neighbour it never regrows and its grass is dessicated."""
  o = oracle.Oracle(commons_pack, util.world_seed(2)); o.reset()
  t = o.tables
  st = t["ch_states"]; s_apple, s_wait = int(st[0]), int(st[1])
  wait_k = [int(x) for x in st[4:]]
  s_grass, s_dess = int(st[2]), int(st[3])
  lay = t["state_layer"]
  live_l, wait_l, bg_l = int(lay[s_apple]), int(lay[s_wait]), int(lay[s_grass])
  disc = t["disc_offsets"].reshape(-1, 2)
  cells = t["apple_cells"]
  rng = np.random.default_rng(1)
  prev_live = None
  seen_dess = False
  total = 0.0
  for s in range(400):
    o.step(rng.choice(8, size=16, p=np.array([0, 8, 3, 2, 3, 2, 2, 0]) / 20.0).astype(np.int32))
    total += o.rewards().sum()
    grid = o.dump()[0]
    live = grid[live_l] == s_apple
    if prev_live is not None:
      for c in cells:
        y, x = divmod(int(c), o.W)
        w = int(grid[wait_l, y, x])
        if w in wait_k and not live[y, x]:
          k = sum(bool(prev_live[y + dy, x + dx]) for dx, dy in disc
                  if 0 <= y + dy < o.H and 0 <= x + dx < o.W)
          # state set this frame from the count at the end of the previous frame
          assert wait_k.index(w) == k, (s, x, y)
          assert int(grid[bg_l, y, x]) == (s_dess if k == 0 else s_grass)
          seen_dess |= k == 0
    prev_live = live
  assert total > 100 and seen_dess


@pytest.mark.skipif(not os.path.isdir(refshim.DEFAULT_REFERENCE_ROOT),
                    reason="reference tree not present (GPU box)")
def test_lowering_refuses_components_it_does_not_implement(commons_closed_pack):
  settings, mod, _ = refshim.build_settings("commons_harvest__closed", ("default",) * 16)
  assert pack.dumps(lower.lower("x", settings, mod.ACTION_SET)) == commons_closed_pack
  # a component the engine has no rule for is refused by name ...
  import copy
  alien = copy.deepcopy(dict(settings))
  alien["simulation"]["scene"]["components"].append(
      {"component": "HiddenAgendaVoting", "kwargs": {}})
  with pytest.raises(NotImplementedError, match="HiddenAgendaVoting"):
    lower.lower("x", alien, mod.ACTION_SET)
  # ... Role / RoleBasedRewardTile only while they are inert: the partnership map
  # lowers with the default roles, not with a rewarded role on an avatar
  settings, mod, _ = refshim.build_settings("commons_harvest__partnership", ("default",) * 7)
  lower.lower("x", settings, mod.ACTION_SET)
  rewarded = copy.deepcopy(dict(settings))
  for obj in rewarded["simulation"]["gameObjects"]:
    for comp in obj["components"]:
      if comp["component"] == "Role":
        comp["kwargs"]["role"] = "putative_cooperator"
  with pytest.raises(NotImplementedError, match="RoleBasedRewardTile"):
    lower.lower("x", rewarded, mod.ACTION_SET)
  # a level without a lowering is refused outright
  bogus = dict(settings, levelName="hidden_agenda")
  with pytest.raises(NotImplementedError):
    lower.lower("x", bogus, mod.ACTION_SET)


# ---------------------------------------------------------------- territory__rooms

@pytest.mark.skipif(not os.path.isdir(refshim.DEFAULT_REFERENCE_ROOT),
                    reason="reference tree not present (GPU box)")
def test_committed_territory_pack_is_what_the_reference_config_lowers_to(territory_pack):
  import sys
  settings, _, _ = refshim.build_settings("territory__rooms", ("default",) * 9)
  # territory__rooms re-uses its base config's action table (territory.py:592-602)
  action_set = sys.modules["meltingpot.configs.substrates.territory"].ACTION_SET
  blob = pack.dumps(lower.lower("territory__rooms", settings, action_set))
  assert blob == territory_pack, "run tools/make_packs.py"


def test_territory_names_the_stacks_worth_pre_blending(territory_pack):
  """`composite_hints` (mp_create's composite cache takes them first): a claimed
  resource that pays is texture + wet paint + dry paint of ONE player
  (territory.py:356-507; Resource / RewardIndicator in components.lua) — and a
  rollout shows exactly those triples, late in an episode on most resources."""
  t = pack.loads(territory_pack)
  names = bytes(t["state_names"]).split(b"\0")
  hints = t["composite_hints"].reshape(-1, 3)
  assert len(hints) == 9
  for i, (a, b, c) in enumerate(hints):
    assert names[a] == b"resource_texture.unclaimed"
    assert names[b] == f"resource.claimed_by_{i + 1}".encode()
    assert names[c] == f"reward_indicator.dry_claimed_by_{i + 1}".encode()
  o = oracle.Oracle(territory_pack, util.world_seed(0)); o.reset()
  rng = np.random.default_rng(0)
  for _ in range(300):
    o.step(rng.integers(0, 9, size=9).astype(np.int32))
  grid = o.dump()[0]
  layer = t["state_layer"]
  wanted = {tuple(h) for h in hints.tolist()}
  paying = 0
  for cell in t["resource_cells"]:
    y, x = divmod(int(cell), grid.shape[2])
    col = [int(s) for s in grid[:, y, x] if s]
    dry = [s for s in col if names[s].startswith(b"reward_indicator.dry")]
    if dry:
      stack = tuple(s for s in col if names[s].split(b".")[0] in
                    (b"resource_texture", b"resource", b"reward_indicator"))
      assert tuple(sorted(stack, key=lambda s: layer[s])) in wanted, [names[s] for s in stack]
      paying += 1
  assert paying >= 20
  o.close()


def test_territory_pack_constants(territory_pack):
  t = pack.loads(territory_pack)
  hdr = t["hdr"]
  # territory__rooms.py:40-62 (21x21, TORUS :91), 9 players (:102), 9 actions
  assert (hdr[lower.HDR_H], hdr[lower.HDR_W]) == (21, 21)
  assert hdr[lower.HDR_P] == 9 and hdr[lower.HDR_NACT] == 9
  assert len(t["resource_cells"]) == 180
  i32, f64, thr = t["tr_i32"], t["tr_f64"], t["tr_thr"]
  # Resource kwargs (territory.py:404-413)
  assert tuple(i32[:3]) == (2, 25, 15)                    # health, rewardDelay, repair delay
  assert tuple(f64[:3]) == (1.0, 0.01, 0.1)               # reward, rewardRate, self repair p
  assert thr[0] == lower.prob_threshold(0.01) and thr[1] == lower.prob_threshold(0.1)
  # ResourceClaimer beam (territory.py:731-738): length 2, radius 0
  assert tuple(i32[3:5]) == (2, 0)
  # GraduatedSanctionsMarking (territory.py:802-818): recovery 50, two levels:
  # level 1 hit -> +1, freeze 25; level 2 hit -> -1, removed
  assert i32[6] == 50 and i32[7] == 2
  assert tuple(i32[10:16]) == (1, 25, 0, -1, 0, 1)
  # 1 zap + 9 brush + 9 claim hits
  assert len(t["tr_hits"]) == 19 and len(set(t["tr_hits"].tolist())) == 19


def _territory_tables(o):
  t = o.tables
  st = [int(x) for x in t["tr_states"]]
  P = o.P
  return dict(unclaimed=st[0], destroyed=st[1], claimed=st[10:10 + P],
              res_layer=int(t["state_layer"][st[0]]), cells=t["resource_cells"])


def test_territory_resource_rules(territory_pack):
  """territory/components.lua:51-210: resources start unclaimed with health 2; the
  paintbrush claims the wall an avatar faces; a destroyed resource never comes
  back; rewards only flow to owners of claimed resources."""
  o = oracle.Oracle(territory_pack, util.world_seed(5)); o.reset()
  k = _territory_tables(o)
  grid, avat, glob = o.dump()
  res0 = np.array([grid[k["res_layer"]].flat[c] for c in k["cells"]])
  assert glob[5] == 2 * 180                     # sum of health
  # the reset frame already ran the updaters: brushes may have claimed a wall
  assert set(res0.tolist()) <= {k["unclaimed"], *k["claimed"]}
  rng = np.random.default_rng(2)
  destroyed_ever = np.zeros(180, bool)
  total = np.zeros(9)
  seen_claim = seen_destroy = False
  prev_owners = set()
  for s in range(1200):
    o.step(rng.choice(9, size=9, p=np.array([0, 5, 1, 1, 1, 2, 2, 6, 2]) / 20.0).astype(np.int32))
    r = o.rewards()
    grid, avat, glob = o.dump()
    res = np.array([grid[k["res_layer"]].flat[c] for c in k["cells"]])
    # the destroyed state has no layer: the cell is free for avatars from then on
    is_destroyed = ~np.isin(res, [k["unclaimed"], *k["claimed"]])
    assert not (destroyed_ever & ~is_destroyed).any()      # destroyed is forever
    destroyed_ever |= is_destroyed
    owners = {k["claimed"].index(int(x)) for x in res if int(x) in k["claimed"]}
    # resource rewards (+1.0) only reach whoever owned a resource when the frame
    # began; zaps and sanctions carry no reward in this substrate
    for p in range(9):
      assert r[p] >= 0 and (r[p] == 0 or p in prev_owners)
    prev_owners = owners
    total += r
    assert glob[3] == sum(int(x) in k["claimed"] for x in res)
    assert 180 <= glob[5] <= 360   # health 1 or 2 (it resets on destruction, :164)
    seen_claim |= bool(owners)
    seen_destroy |= bool(is_destroyed.any())
  assert seen_claim and seen_destroy and total.sum() > 0


def test_territory_graduated_sanctions(territory_pack):
  """avatar_library.lua:948-1121 with territory.py:802-818: the first zap freezes
  the victim for 25 frames at level 2, a second one removes it for good."""
  o = oracle.Oracle(territory_pack, util.world_seed(7)); o.reset()
  rng = np.random.default_rng(4)
  levels_seen, removed = set(), False
  frozen_frames = 0
  for s in range(1500):
    o.step(rng.choice(9, size=9, p=np.array([0, 6, 1, 1, 1, 3, 3, 5, 0]) / 20.0).astype(np.int32))
    _, avat, _ = o.dump()
    for p in range(9):
      extra = int(avat[p, 7])
      level, freeze = extra & 15, (extra >> 4) & 255
      levels_seen.add(level)
      assert level in (1, 2) and freeze <= 25
      frozen_frames += freeze > 0
      removed |= avat[p, 3] == 0
  assert levels_seen == {1, 2} and frozen_frames > 0 and removed


@pytest.mark.skipif(not os.path.isdir(refshim.DEFAULT_REFERENCE_ROOT),
                    reason="reference tree not present (GPU box)")
def test_committed_territory_open_pack_is_what_the_reference_config_lowers_to(territory_open_pack):
  import sys
  settings, _, _ = refshim.build_settings("territory__open", ("default",) * 9)
  action_set = sys.modules["meltingpot.configs.substrates.territory"].ACTION_SET
  blob = pack.dumps(lower.lower("territory__open", settings, action_set))
  assert blob == territory_open_pack, "run tools/make_packs.py"
  hdr = pack.loads(blob)["hdr"]
  # territory__open.py:45-70: 23 x 39, BOUNDED (topology 0), 9 players
  assert (hdr[lower.HDR_H], hdr[lower.HDR_W], hdr[lower.HDR_P]) == (23, 39, 9)
  # territory__inside_out places optional resources and spawn points with
  # 'choice' map characters, drawn once per episode (prefab_utils.lua:101-103)
  settings, _, _ = refshim.build_settings("territory__inside_out", ("default",) * 5)
  t = pack.loads(pack.dumps(lower.lower("territory__inside_out", settings, action_set)))
  assert len(t["choice_n"]) == 48 + 64 + 16            # 'A', 'B' and 'Q' cells
  assert sorted(set(t["choice_n"].tolist())) == [3, 4, 7]  # odds 2:1, 1:3, 1:6
  assert len(t["resource_cells"]) == 88 + 48 + 64       # 'R' + optional ones


# ---------------------------------------------------------------- coins

@pytest.mark.skipif(not os.path.isdir(refshim.DEFAULT_REFERENCE_ROOT),
                    reason="reference tree not present (GPU box)")
def test_committed_coins_pack_is_what_the_reference_config_lowers_to(coins_pack):
  """coins.py draws the map size and the two coin colours with Python's `random`
  inside build(): the committed pack holds every map the generator can draw (one
  per-world choice of 36 outcomes) and every colour's coin and avatar states (one
  per-world draw of the 20 ordered colour pairs)."""
  import random
  random.seed(0)
  settings, mod, config = refshim.build_settings("coins", ("default",) * 2)
  settings = lower.coins_with_every_map(settings, mod, config)
  blob = pack.dumps(lower.lower("coins", settings, mod.ACTION_SET))
  assert blob == coins_pack, "run tools/make_packs.py"
  t = pack.loads(blob)
  assert t["choice_n"].tolist() == [-36] and "object_choice_hi" in t
  hdr = t["hdr"]
  # padded to max_width + 2 x max_height + 2 (coins.py:45-84, WORLD.RGB 136 x 136)
  assert (hdr[lower.HDR_H], hdr[lower.HDR_W], hdr[lower.HDR_P]) == (17, 17, 2)
  assert hdr[lower.HDR_NACT] == 7 and hdr[lower.HDR_NHITS] == 0   # no beams at all
  # rewards: self +1 either way, the other player 0 / -2 (coins.py:396-403)
  assert t["co_f64"][:8].tolist() == [1.0, 1.0, 0.0, -2.0] * 2
  assert t["co_thr"][0] == lower.prob_threshold(0.0005)
  assert sorted(t["co_i32"][:2].tolist()) == [0, 1]          # one colour each


def test_every_coins_world_has_its_own_map(coins_pack):
  """coins.py:45-82,500: width and height are drawn (10..15 each) when an
  environment is BUILT, so N environments are N maps and an environment keeps its
  map through its episodes.  Here: one per-world choice of 36 outcomes."""
  seen = {}
  for w in range(300):
    o = oracle.Oracle(coins_pack, util.world_seed(w)); o.reset()
    grid = o.dump()[0]
    occupied = (grid != 0).any(axis=0)
    ys, xs = np.where(occupied)
    size = (int(xs.max()) - 1, int(ys.max()) - 1)       # interior width, height
    assert xs.min() == 0 and ys.min() == 0 and 10 <= size[0] <= 15 and 10 <= size[1] <= 15
    # the ring of walls of that size, and nothing outside it
    ring = np.zeros_like(occupied)
    ring[0, :size[0] + 2] = ring[size[1] + 1, :size[0] + 2] = True
    ring[:size[1] + 2, 0] = ring[:size[1] + 2, size[0] + 1] = True
    assert occupied[ring].all() and not occupied[size[1] + 2:].any() and not occupied[:, size[0] + 2:].any()
    # the two avatars stand on the map's two spawn points (coins.py:63-68)
    _, avat, _ = o.dump()
    assert sorted(map(tuple, avat[:, :2].tolist())) == sorted([(size[0] - 1, 2), (2, size[1] - 1)])
    seen[size] = seen.get(size, 0) + 1
    if w < 20:                                         # the map survives the episodes
      for _ in range(3):
        o.reset()
        assert np.array_equal((o.dump()[0] != 0).any(axis=0) | _avatar_cells(o), occupied | _avatar_cells(o))
    o.close()
  assert len(seen) == 36 and min(seen.values()) >= 1   # every (width, height) occurs


def _coins_colours(t, grid, avat):
  """(colour of player 1, colour of player 2, colours of the coins lying about)."""
  alive = t["co_colour_alive"].reshape(2, 5).tolist()
  coin = t["co_colour_coin"].tolist()
  mine = []
  for p in range(2):
    here = [alive[p].index(int(v)) for v in grid[:, avat[p, 1], avat[p, 0]] if int(v) in alive[p]]
    assert len(here) == 1
    mine.append(here[0])
  lying = {coin.index(int(v)) for v in np.unique(grid) if int(v) in coin}
  return mine[0], mine[1], lying


def test_every_coins_world_has_its_own_colours(coins_pack):
  """coins.py:500-514: build() samples two of the five colours; player 1's avatar
  and coins get the first, player 2's the second.  Here: one per-world draw of the
  20 ordered pairs, kept through the world's episodes like its map."""
  t = pack.loads(coins_pack)
  assert t["co_colour_coin"].shape == (5,) and t["co_colour_alive"].size == 10
  seen = set()
  for w in range(300):
    o = oracle.Oracle(coins_pack, util.world_seed(w)); o.reset()
    a, b, lying = _coins_colours(t, *o.dump()[:2])
    assert a != b and lying <= {a, b}
    seen.add((a, b))
    if w < 10:
      rng = np.random.default_rng(w)
      for _ in range(2):
        while not o.done:
          o.step(rng.integers(0, 7, size=2).astype(np.int32))
        a2, b2, lying = _coins_colours(t, *o.dump()[:2])
        assert (a2, b2) == (a, b) and lying <= {a, b} and len(lying) == 2
        o.reset()
      # the avatar's sprite is its colour's: the two agents' views differ from a
      # world of another pair only in palette, so at least the world frame differs
      assert o.render_world().any()
    o.close()
  assert len(seen) == 20


def _avatar_cells(o):
  grid, avat, _ = o.dump()
  m = np.zeros(grid.shape[1:], bool)
  for x, y in avat[:, :2]:
    m[y, x] = True
  return m


def test_coins_rules(coins_pack):
  """Coin:onEnter / ChoiceCoinRegrow / PartnerTracker (coins/components.lua): a
  coin of one's own colour pays +1 and nobody else; a mismatched one pays +1 and
  costs the partner 2, who sees MISMATCHED_COIN_COLLECTED_BY_PARTNER that frame."""
  o = oracle.Oracle(coins_pack, util.world_seed(3)); o.reset()
  assert [e[0] for e in o.events()] == [9, 9]
  ptype = [int(x) for x in o.tables["co_i32"][:2]]
  rng = np.random.default_rng(1)
  seen_match = seen_mismatch = False
  live_prev = 0
  while not o.done:
    o.step(rng.choice(7, size=2, p=np.array([0, 8, 2, 2, 2, 1, 1]) / 16).astype(np.int32))
    r, flags = o.rewards(), o.num_others_cleaned()
    want_r, want_f = np.zeros(2), np.zeros(2)
    for typ, player, packed in o.events():
      assert typ == 10
      p, coin, mine = player - 1, packed & 1, packed >> 1
      assert mine == ptype[p]
      want_r[p] += 1.0
      if coin != mine:
        want_r[1 - p] += -2.0
        want_f[1 - p] = 1.0
        seen_mismatch = True
      else:
        seen_match = True
    assert np.array_equal(r, want_r) and np.array_equal(flags, want_f)
    live = int(o.dump()[2][3])
    assert live >= live_prev - len(o.events())   # coins only leave by being collected
    live_prev = live
  assert seen_match and seen_mismatch


def test_a_step_on_a_finished_world_reports_nothing(coins_pack):
  """World 835 of the batch the deep soak runs (tests/tools/deep_soak.py, round 6): its episode
  ends by the interval draw on the very step that pays a coin.  A step asked of it afterwards
  (no reset: MpConfig.auto_reset = 0) moves nothing and reports neither that reward nor that
  event again — the frozen world of csrc/step_common.h dispatch()."""
  w = 835
  o = oracle.Oracle(coins_pack, util.world_seed(w)); o.reset()
  s = 0
  while not o.done:
    o.step(util.hashed_actions([w], s, o.P, num_actions=7)[0])
    s += 1
  assert s == 499 and o.rewards().tolist() == [0.0, 1.0] and o.events() == [(10, 2, 3)]
  before = [x.copy() for x in o.dump()]
  for _ in range(2):
    o.step(util.hashed_actions([w], s, o.P, num_actions=7)[0])
    assert o.done and o.rewards().tolist() == [0.0, 0.0] and o.events() == []
    assert all(np.array_equal(a, b) for a, b in zip(before, o.dump()))


def test_territory_inside_out_maps_vary_per_episode():
  """`choice` map characters (prefab_utils.lua:101-103): odds 2:1 for 'A'
  resources, 1:3 for 'B', 1:6 for 'Q' spawn points (territory__inside_out.py:72-85),
  redrawn at every world build."""
  from meltingpot_amd import engine
  b = engine.load_pack("territory__inside_out")
  t = pack.loads(b)
  counts = []
  for w in range(60):
    o = oracle.Oracle(b, util.world_seed(w)); o.reset()
    counts.append(int(o.dump()[2][5]) // int(t["tr_i32"][0]))   # sum of health / initial health
    o.close()
  # 88 'R' always + 48 'A' at 2/3 + 64 'B' at 1/4: mean 136, sd 4.8
  assert 88 < min(counts) and max(counts) < 200 and len(set(counts)) > 8
  assert abs(np.mean(counts) - 136.0) < 2.5


# ---------------------------------------------------------------- A10s: the serial generator


def test_mt19937_64_known_answers():
  """oracle/mt19937_64.h against the C++ standard's known answer ([rand.predef]: the
  10000th output of a default-constructed std::mt19937_64 is 9981545732273789042)
  and the generator's published first outputs for seed 5489."""
  from oracle import oracle
  L = oracle.lib()
  assert L.orc_mt19937_64(5489, 10000) == 9981545732273789042
  assert [L.orc_mt19937_64(5489, n) for n in (1, 2, 3)] == [
      14514284786278117030, 4620546740167642908, 13109570281517897720]
  # conversions (mt19937_64.h): uniformReal = round-to-nearest-even of x / 2^11, below 2^53
  x = L.orc_mt19937_64(5489, 1)
  u = L.orc_mt19937_64_draw(5489, 0, 0, 0)
  assert u == min(int(round(x / 2048.0)), (1 << 53) - 1) and u < (1 << 53)
  assert float(u) * 2.0 ** -53 == float(np.float64(x) / np.float64(2.0 ** 64))
  # uniformInt(0, n - 1): Lemire's (x * n) >> 64 without a rejection for this draw, and
  # the scaling form x / (max / n)
  for n in (2, 3, 4, 7, 122, 147):
    assert L.orc_mt19937_64_draw(5489, 0, 1, n) == (x * n) >> 64
    scaling = 0xFFFFFFFFFFFFFFFF // n
    assert x < n * scaling and L.orc_mt19937_64_draw(5489, 0, 2, n) == x // scaling
  assert L.orc_mt19937_64_draw(5489, 0, 1, 1) == 0


def test_serial_generator_mode_runs_the_same_rules(clean_up_pack):
  """A10s: with ONE mt19937_64 per world consumed in call order the episode is another
  sample of the same rules — deterministic per seed, different from the counter-based
  run, reseeded per episode (api_factory.lua:89), with the same invariants (7 avatars
  alive on distinct cells after the reset, dirt spawning after its delay)."""
  from oracle import oracle
  def run(serial, seed, method=0, back=0, steps=120):
    o = oracle.Oracle(clean_up_pack, seed)
    o.set_option("A10s_serial_mt19937", serial)
    o.set_option("A10s_int_method", method)
    o.set_option("A10s_shuffle_back", back)
    o.reset()
    first = o.dump()
    rng = np.random.default_rng(1)
    for a in rng.integers(0, 9, size=(steps, 7), dtype=np.int32):
      o.step(a)
    return first, o.dump(), o
  f0, e0, _ = run(0, 11)
  f1, e1, o1 = run(1, 11)
  f1b, e1b, _ = run(1, 11)
  assert np.array_equal(f1[1], f1b[1]) and np.array_equal(e1[0], e1b[0])    # deterministic
  assert not np.array_equal(f0[1], f1[1])       # another sample than the counter-based one
  assert not np.array_equal(run(1, 12)[0][1], f1[1])                         # seed-sensitive
  alive = f1[1][:, 3] == 1
  assert alive.all() and len({(int(x), int(y)) for x, y in f1[1][:, :2]}) == 7
  assert int(e1[2][3]) > 0, "dirt has spawned by step 120"
  # the conversions are assumptions of their own: each changes the sample, not the rules
  # (deterministic too; whether a given run can tell them apart depends on its conflicts)
  for kw in ({"method": 1}, {"back": 1}):
    a, b = run(1, 11, **kw), run(1, 11, **kw)
    assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1])
  # a second episode reseeds: not a replay of the first
  o1.reset()
  assert not np.array_equal(o1.dump()[1], f1[1])
