"""externality_mushrooms__dense (lua/levels/externality_mushrooms/components.lua,
configs/substrates/externality_mushrooms.py + externality_mushrooms__dense.py): the committed
pack against the reference config, and the oracle's restatement of the rules — an independent
Python model of who a mushroom pays, how long it lives and which sites may grow, run next to
oracle rollouts on a map that starts full of mushrooms; the invariant the HIP kernel builds on
(the Lua's set of potential sites = the mushrooms that waited when the frame began, its counter
that size minus the mushrooms the map starts with); scripted meals with hand-computed
expectations."""
import os

import numpy as np
import pytest

import util
from meltingpot_amd import engine, lower, pack, refshim
from oracle import oracle

HAVE_REFERENCE = os.path.isdir("/root/reference/meltingpot")
EAT, ZAP, SANCTION, REMOVAL, SET_LEVEL = 20, 1, 6, 7, 8
ZAP_HEAVY = [1, 4, 1, 1, 1, 2, 2, 5]   # weights over the ACTION_SET: walks forward, zaps often
NAME = "externality_mushrooms__dense"
TOTAL = (1.0, 2.0, 3.0, -1.0)          # totalReward per type (externality_mushrooms.py:578-583)
DIGEST = (0, 10, 15, 15)
PERISH = (200, 100, 75, None)
SPORES = (3, 3, 3, 1)


@pytest.fixture(scope="module")
def mushroom_pack() -> bytes:
  return engine.load_pack(NAME)


def lush(pack_bytes, frac=0.45, seed=0, grow=0.35, types=(0.4, 0.25, 0.2, 0.15)):
  """The pack with a share of its sites starting as live mushrooms (of types drawn with
  `types`) and every spore growing each type with probability `grow`: random play meets
  meals, digestion, spores, destruction and perishing from the first steps (the stock map's
  ten mushrooms are gone after 200 frames of it)."""
  t = pack.loads(pack_bytes)
  rng = np.random.default_rng(seed)
  W = int(t["hdr"][lower.HDR_W])
  objs = t["objects"].reshape(-1, 4).copy()
  H, L = int(t["hdr"][lower.HDR_H]), int(t["hdr"][lower.HDR_L])
  grid = t["init_grid"].reshape(L, H, W).copy()
  st = [int(s) for s in t["em_states"][:5]]
  layer = int(t["state_layer"][st[0]])
  live = 0
  for row in objs:
    if row[0] != lower.KIND_MUSHROOM:
      continue
    s = st[4]
    if rng.random() < frac:
      s = st[int(rng.choice(4, p=np.asarray(types) / sum(types)))]
      live += 1
    row[3] = s
    grid[layer, row[2], row[1]] = 0 if s == st[4] else s
  ci = t["em_i32"].copy()
  ci[7] = live
  thr = t["em_thr"].copy()
  thr[:16] = lower.prob_threshold(grow)
  return util.patch_pack(pack_bytes, tables={"objects": objs.reshape(-1), "init_grid": grid.reshape(-1),
                                             "em_i32": ci, "em_thr": thr})


def unpack_avatar(v):
  """avat[p][7] of the dump -> level, freeze, removal, noZap, movementAllowed, disallowZapping."""
  return v & 15, (v >> 4) & 255, (v >> 12) & 15, (v >> 16) & 255, (v >> 24) & 1, (v >> 25) & 1


@pytest.mark.skipif(not HAVE_REFERENCE, reason="reference tree not present (GPU box)")
def test_committed_pack_is_what_the_reference_config_lowers_to(mushroom_pack):
  import sys
  settings, mod, config = refshim.build_settings(NAME, ("default",) * 5)
  action_set = sys.modules["meltingpot.configs.substrates.externality_mushrooms"].ACTION_SET
  assert pack.dumps(lower.lower(NAME, settings, action_set)) == mushroom_pack, "run tools/make_packs.py"
  t = pack.loads(mushroom_pack)
  hdr = t["hdr"]
  assert (hdr[lower.HDR_H], hdr[lower.HDR_W], hdr[lower.HDR_P]) == (14, 23, 5)   # ..._dense.py:29-44, :84
  assert hdr[lower.HDR_MAXFRAMES] == 5000                                        # externality_mushrooms.py:1063
  rows = [r for r in mod.ASCII_MAP.strip("\n").split("\n")]
  assert len(t["mushroom_cells"]) == sum(r.count(c) for r in rows for c in " RGBO") == 231
  names = bytes(t["state_names"]).split(b"\0")
  assert [names[s] for s in t["em_states"]] == [
      b"mushroom." + n.encode() for n in lower.MUSHROOM_TYPES] + [
      b"mushroom.wait", b"avatar_marking.level_1", b"avatar_marking.level_2",
      b"avatar_marking.avatarMarkingWait"]
  ci = list(t["em_i32"])
  # minPotentialMushrooms 1 (:752), health 1 (:614), recovery 50, two levels (:992-1003), episode
  # end (:760-762), hit 0, ten mushrooms on the map; spores, digestion, perishing, destruction
  # per type (:585-626); level 1: +1, freeze 25; level 2: -1, remove
  assert ci[:8] == [1, 1, 50, 2, 1000, 100, 0, sum(r.count(c) for r in rows for c in "RGBO")]
  assert ci[8:24] == list(SPORES) + list(DIGEST) + [200, 100, 75, 1 << 30] + [-1, -1, -1, 0]
  assert ci[24:30] == [1, 25, 0, -1, 0, 1]
  assert list(t["em_f64"]) == list(TOTAL) + [0.0] * 4
  probs = [0.25, 0, 0, 0, 0.25, 0.4, 0, 0, 0.25, 0.4, 0.6, 0, 0, 0, 0, 1.0]      # :728-751
  assert list(t["em_thr"]) == [lower.prob_threshold(p) for p in probs + [0, 0, 0, 0.25] + [0.2]]
  assert list(t["zapper_i32"]) == [3, 3, 1, 50, 0] and list(t["zapper_f64"]) == [0.0, 0.0]   # :854-865
  assert bytes(t["hit_names"]) == b"zapHit\0"
  assert bytes(t["action_names"]) == b"move\0turn\0fireZap\0"


def _site_states(t, grid):
  st = [int(s) for s in t["em_states"][:4]]
  layer = int(t["state_layer"][st[0]])
  s = grid[layer].reshape(-1)[t["mushroom_cells"]]
  return np.where(s == 0, -1, s.astype(int) - st[0])   # -1 waiting, else the type


def test_rules_hold_over_rollouts_on_a_lush_map(mushroom_pack):
  """Every frame of five rollouts: the set of potential sites the oracle keeps (the Lua's, flag
  by flag) is the sites that waited when the frame began, and its counter is that number minus
  the mushrooms the map starts with — what step_mushroom.h derives instead of keeping; an
  `eating_mushroom` event pays by its type's rule (everybody alive; the eater; the others); a
  mushroom is gone `delay` frames after it appeared; a site only changes between waiting and
  live, never from type to type; digestion freezes the eater; READY_TO_SHOOT follows the
  cooldown of 3 and the zap prevention of a sanction."""
  eats = np.zeros(4, int)
  perished = grown = destroyed_by_eating = 0
  for seed in range(5):
    pk = lush(mushroom_pack, seed=seed)
    t = pack.loads(pk)
    n_live0 = int(t["em_i32"][7])
    P = 5 if seed < 3 else 3
    o = oracle.Oracle(pk, util.world_seed(seed), P); o.reset()
    rng = np.random.default_rng(seed)
    grid, avat, glob = o.dump()
    sites = _site_states(t, grid)
    assert (sites >= 0).sum() == n_live0 == glob[3]
    assert glob[7] == (sites < 0).sum() and glob[6] - 1000 == glob[7] - n_live0
    age = np.where(sites >= 0, 1, 0)
    for step in range(400):
      acts = rng.choice(8, size=P, p=np.array(ZAP_HEAVY) / sum(ZAP_HEAVY)).astype(np.int32)
      alive_before = avat[:, 3].copy()
      waiting_before = (sites < 0).sum()
      o.step(acts)
      grid, avat, glob = o.dump()
      new = _site_states(t, grid)
      # the potential sites of this frame: those that waited when it began
      assert glob[7] == waiting_before and glob[6] - 1000 == waiting_before - n_live0
      ev = o.events()
      eaten = [(a - 1, b - 1) for ty, a, b in ev if ty == EAT]
      if (avat[:, 3] == alive_before).all() and not any(ty == REMOVAL for ty, _, _ in ev):
        # (nobody came or went this frame: the living are the living throughout)
        expect = np.zeros(P)
        for p, ty in eaten:
          for q in range(P):
            if not avat[q, 3]:
              continue
            if ty == 0:
              expect[q] += TOTAL[0] if q == p else 0.0
            elif ty == 2:
              expect[q] += 0.0 if q == p else TOTAL[2] / (P - 1)
            else:
              expect[q] += TOTAL[ty] / P
        assert np.allclose(o.rewards(), expect, rtol=0, atol=1e-12), (step, eaten, o.rewards(), expect)
      for p, ty in eaten:
        eats[ty] += 1
        level, freeze, removal, nozap, allowed, disallow = unpack_avatar(int(avat[p, 7]))
        if DIGEST[ty] and not any(tt == SANCTION and b - 1 == p for tt, _, b in ev):
          assert (freeze, allowed) == (DIGEST[ty], 0)
      # sites: waiting <-> live only; ages; perishing
      changed = new != sites
      assert not ((sites >= 0) & (new >= 0) & changed).any()
      grown += ((sites < 0) & (new >= 0)).sum()
      for i in np.flatnonzero((sites >= 0) & (new < 0)):
        d = PERISH[sites[i]]
        if d is not None and age[i] == d:
          perished += 1
      age = np.where(new >= 0, np.where(changed, 1, age + 1), 0)
      for ty in range(3):
        assert (age[new == ty] <= PERISH[ty]).all()
      if any(ty == 3 for _, ty in eaten):
        destroyed_by_eating += ((sites == 0) & (new < 0)).sum()
      sites = new
      assert glob[3] == (sites >= 0).sum()
      for p in range(P):
        level, freeze, removal, nozap, allowed, disallow = unpack_avatar(int(avat[p, 7]))
        assert level in (1, 2) and (freeze > 0) <= (allowed == 0) and (nozap > 0) == bool(disallow)
        want = max(0.0, 1.0 - avat[p, 4] / 3.0) if avat[p, 3] else 0.0
        assert o.ready_to_shoot()[p] == want
  assert (eats >= 5).all() and perished > 50 and grown > 100 and destroyed_by_eating > 5, (
      eats, perished, grown, destroyed_by_eating)


def test_sanctions_removals_and_returns(mushroom_pack):
  """Zap-heavy play on the stock pack: a first hit freezes the target for 25 frames at level 2
  (`set_sanctioning_level`), a second one within 50 frames removes it a frame later
  (`removal_due_to_sanctioning`), 50 frames after that it is back with its marking at the
  level it left with, announced by another `set_sanctioning_level`; an untouched level 2
  recovers after 50 frames."""
  P = 5
  removed = returned = recovered = frozen = 0
  for seed in range(4):
    o = oracle.Oracle(mushroom_pack, util.world_seed(10 + seed), P); o.reset()
    rng = np.random.default_rng(seed)
    dead_for = np.zeros(P, int)
    grid, avat, glob = o.dump()
    for step in range(1200):
      acts = rng.choice(8, size=P, p=np.array(ZAP_HEAVY) / sum(ZAP_HEAVY)).astype(np.int32)
      before = avat.copy()
      o.step(acts)
      grid, avat, glob = o.dump()
      ev = o.events()
      for ty, a, b in ev:
        if ty == REMOVAL:
          level, freeze, removal, nozap, allowed, disallow = unpack_avatar(int(avat[b - 1, 7]))
          assert (level, removal, freeze, allowed) == (1, 1, 1, 0) and avat[b - 1, 3] == 1
          removed += 1
      for p in range(P):
        level, freeze, removal, nozap, allowed, disallow = unpack_avatar(int(avat[p, 7]))
        b_level, b_freeze, b_removal, *_ = unpack_avatar(int(before[p, 7]))
        if before[p, 3] and not avat[p, 3]:
          assert b_removal == 1          # gone exactly one frame after the removing hit
          dead_for[p] = 0
        elif not before[p, 3] and avat[p, 3]:
          assert dead_for[p] == 50 and (8, p + 1, level) in ev
          returned += 1
        if not avat[p, 3]:
          dead_for[p] += 1
        if (SET_LEVEL, p + 1, 2) in ev and avat[p, 3] and before[p, 3] and b_level == 1 and not any(
            tt == REMOVAL and bb == p + 1 for tt, _, bb in ev):   # (two hits in one frame remove at once)
          assert (freeze, allowed, nozap, disallow) == (25, 0, 25, 1)
          frozen += 1
        if (SET_LEVEL, p + 1, 1) in ev and before[p, 3] and b_level == 2 and not any(
            tt == SANCTION and bb == p + 1 for tt, _, bb in ev):
          recovered += 1
  assert removed > 20 and returned > 15 and frozen > 40 and recovered > 10, (
      removed, returned, frozen, recovered)


def test_scripted_meals(mushroom_pack):
  """Hand-computed: on a map whose sites all start as type-2 mushrooms (half for the eater, half
  for the others: 2 / 5 each to all five), a step forward onto one pays everybody 0.4 and
  freezes the eater for ten frames; on a map of type-3 mushrooms (nothing for the eater) a
  meal pays the four others 0.75 each; a zap destroys the three mushrooms ahead and passes."""
  t0 = pack.loads(mushroom_pack)
  W = int(t0["hdr"][lower.HDR_W])
  for ty, expect, digest in ((1, [0.4] * 5, 10), (2, [0.0] + [0.75] * 4, 15), (0, [1.0] + [0.0] * 4, 0),
                             (3, [-0.2] * 5, 15)):
    types = [0.0] * 4
    types[ty] = 1.0
    pk = lush(mushroom_pack, frac=1.1, types=types, grow=0.0)
    t = pack.loads(pk)
    o = oracle.Oracle(pk, util.world_seed(2), 5); o.reset()
    # avatars on distinct rows of the open field, all facing east; 0 is the one that acts
    for p in range(5):
      assert o.place_avatar(p, 3, 3 + 2 * p, 1)
    grid, avat, glob = o.dump()
    n0 = glob[3]
    o.step(np.array([1, 0, 0, 0, 0], np.int32))            # 0 steps east, onto a mushroom
    assert np.allclose(o.rewards(), expect, rtol=0, atol=1e-15)
    assert (EAT, 1, ty + 1) in o.events()
    grid, avat, glob = o.dump()
    assert glob[3] == n0 - 1    # (type 4 destroys type-1 mushrooms only: none on this map)
    level, freeze, removal, nozap, allowed, disallow = unpack_avatar(int(avat[0, 7]))
    assert (freeze, allowed) == ((digest, 0) if digest else (0, 1))
    o.step(np.array([1, 0, 0, 0, 0], np.int32))            # digesting: it stays where it is
    grid2, avat2, _ = o.dump()
    assert (avat2[0, 0] == avat[0, 0]) == bool(digest)
    if ty == 0:
      # a zap from player 1 (facing east at x = 3, Zapper beamLength 3, beamRadius 1): mushrooms
      # do not stop it
      before = _site_states(t, grid2)
      o.step(np.array([0, 7, 0, 0, 0], np.int32))
      grid3, avat3, _ = o.dump()
      after = _site_states(t, grid3)
      gone = {int(c) for c in t["mushroom_cells"][(before >= 0) & (after < 0)]}
      y = 5
      # (A4's footprint: the centre ray, and from the cell at either side a ray one shorter)
      assert gone == {y * W + x for x in (4, 5, 6)} | {(y + d) * W + x for d in (-1, 1) for x in (3, 4, 5)}
      assert not [e for e in o.events() if e[0] == EAT]


# Worlds (global index = seed) of a 16384-world search (tools/gpu_find_displaced_markings.py,
# tools/history/gpu_r06_call2.sh: 21 of 16384 worlds in 2500 steps) in which a sanctions marking ends up
# on the map AWAY from its living avatar, and the step at which it first does.
DISPLACED = {12246: 104, 14957: 104, 10642: 188}


def stray_markings(o, mark_layer):
  """(cells of on-grid markings no living avatar stands on, living avatars, dead avatars)."""
  grid, avat, _ = o.dump()
  marks = {(int(x), int(y)) for y, x in zip(*np.nonzero(grid[mark_layer]))}
  alive = {(int(a[0]), int(a[1])) for a in avat if a[3]}
  return sorted(marks - alive), sorted(alive), sum(1 for a in avat if not a[3])


def test_the_oracle_reaches_markings_connected_at_a_distance(mushroom_pack):
  """avatar_library.lua:1099-1110 through the oracle's engine, in worlds where it HAPPENS (round 5
  counted these cases in the kernel instead of restating them; round 6 restates them, and this
  is what they look like on the oracle).  World 12246: player 2 loses its marking (it comes
  back from a removal while another marking lies where its own waited), walks on without one,
  and at step 104 resetToInitialLevel's _setLevel (:1010-1026) puts the marking on the map at
  its transform — one cell beside the avatar; from then on the two move as one group, one cell
  apart (A14), and at step 110 player 2's own zap hits its own marking: a `sanctioning` event
  with source == target.  More on-grid markings off their avatars than there are dead avatars
  (whose markings may be orphans) = some living avatar's marking is elsewhere."""
  t = pack.loads(mushroom_pack)
  mark_layer = int(t["state_layer"][int(t["em_states"][5])])
  for w, first in DISPLACED.items():
    o = oracle.Oracle(mushroom_pack, util.world_seed(w), 5)
    o.reset()
    seen = None
    for s in range(first + 12):
      o.step(util.hashed_actions([w], s, 5)[0])
      stray, alive, dead = stray_markings(o, mark_layer)
      if seen is None and len(stray) > dead:
        seen = s
      if w == 12246 and 104 <= s <= 109:
        # the marking that came back beside player 2 stays one cell to its right
        _, avat, _ = o.dump()
        x, y = int(avat[1][0]), int(avat[1][1])
        assert (x + 1, y) in stray, (s, stray, (x, y))
      if w == 12246 and s == 104:
        assert (SET_LEVEL, 2, 1) in o.events()
      if w == 12246 and s == 110:
        assert (SANCTION, 2, 2) in o.events()        # its own beam, its own marking
    assert seen == first, (w, seen)
    o.close()
