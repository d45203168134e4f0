"""coop_mining (lua/levels/coop_mining/components.lua, configs/substrates/coop_mining.py): the
committed pack against the reference config, and the oracle's restatement of the rules —
invariants over rollouts with plentiful ore, and scripted situations with hand-computed
expectations (two miners on one gold ore, the mining window running out, both on one iron)."""
import os

import numpy as np
import pytest

import util
from meltingpot_amd import lower, pack, refshim
from oracle import oracle

HAVE_REFERENCE = os.path.isdir("/root/reference/meltingpot")
MINING, EXTRACTION, PAIR = 13, 14, 15


def rich(pack_bytes, iron=0.02, gold=0.02):
  """The pack with ore growing 100 x faster (the stock rates, 2e-4 / 8e-5 per site and
  frame, leave random play almost nothing to mine)."""
  t = pack.loads(pack_bytes)
  thr = t["cm_thr"].copy()
  thr[0], thr[1] = lower.prob_threshold(iron), lower.prob_threshold(gold)
  return util.patch_pack(pack_bytes, tables={"cm_thr": thr})


@pytest.mark.skipif(not HAVE_REFERENCE, reason="reference tree not present (GPU box)")
def test_committed_pack_is_what_the_reference_config_lowers_to(coop_mining_pack):
  settings, mod, config = refshim.build_settings("coop_mining", ("default",) * 8)
  assert pack.dumps(lower.lower("coop_mining", settings, mod.ACTION_SET, default_players=6)) == (
      coop_mining_pack), "run tools/make_packs.py"
  t = pack.loads(coop_mining_pack)
  hdr = t["hdr"]
  assert (hdr[lower.HDR_H], hdr[lower.HDR_W]) == (27, 27)                   # coop_mining.py:49-77
  assert hdr[lower.HDR_MAXFRAMES] == 5000 and hdr[lower.HDR_DEFAULT_P] == 6  # :486, :471
  assert len(t["ore_cells"]) == mod.ASCII_MAP.count("O")
  names = bytes(t["state_names"]).split(b"\0")
  st = t["cm_states"]
  assert [names[s] for s in st] == [b"ore.oreWait", b"ore.ironRaw", b"ore.goldRaw", b"ore.ironRaw",
                                    b"ore.goldPartial"]
  # cooldown, beam, episode ending, (minNumMiners, miningWindow) of iron and gold (:263-283,:375-388)
  assert list(t["cm_i32"]) == [3, 3, 0, 0, 1000, 100, 1, 2, 2, 3]
  f = t["cm_f64"]
  assert list(f[:4]) == [0.0, 0.0, 1.0, 8.0]            # role "none": mining [0, 0], extracting [1, 8]
  assert list(f[-3:]) == [0.0002, 0.00008, 0.2]
  assert list(t["cm_thr"]) == [lower.prob_threshold(p) for p in (0.0002, 0.00008, 0.2)]
  assert bytes(t["hit_names"]) == b"mine\0" and bytes(t["action_names"]) == b"move\0turn\0mine\0"
  # the two roles build the same substrate
  s2, _, _ = refshim.build_settings("coop_mining", ("target",) * 8)
  assert pack.dumps(lower.lower("coop_mining", s2, mod.ACTION_SET, default_players=6)) == coop_mining_pack


def test_rules_hold_over_rollouts_with_plentiful_ore(coop_mining_pack):
  """Every frame: an iron hit pays the hitter its extraction at once (1.0) and the ore is
  gone; gold pays 8.0 to each of exactly two DIFFERENT miners, with one `extraction_pair`
  event in each direction; nobody is paid without an event; READY_TO_SHOOT follows the
  cooldown of 3; ores only appear on cells no avatar stands on."""
  pk = rich(coop_mining_pack)
  t = pack.loads(pk)
  s_wait, s_iron, s_gold, _, s_part = (int(x) for x in t["cm_states"])
  ore_layer = int(t["state_layer"][s_wait])
  cells = t["ore_cells"]
  rng = np.random.default_rng(2)
  seen = {MINING: 0, EXTRACTION: 0, PAIR: 0}
  gold_extractions = 0
  for seed in range(3):
    o = oracle.Oracle(pk, util.world_seed(seed), 6); o.reset()
    timer = np.zeros(6, int)
    prev_grid = o.dump()[0]
    for step in range(400):
      acts = rng.choice(8, size=6, p=np.array([1, 3, 1, 1, 1, 2, 2, 5]) / 16).astype(np.int32)
      fired = np.zeros(6, bool)
      for p in range(6):                        # MineBeam:update (components.lua:228-244)
        timer[p] = max(timer[p] - 1, 0)
        if acts[p] == 7 and timer[p] == 0:
          timer[p] = 3; fired[p] = True
      o.step(acts)
      grid, avat, glob = o.dump()
      assert np.array_equal(avat[:, 4], timer)
      assert np.allclose(o.ready_to_shoot(), 1.0 - timer / 3.0)
      want = np.zeros(6)
      ev = o.events()
      miners, extracted, pairs = [], [], []
      for typ, a, b in ev:
        seen[typ] += 1
        if typ == MINING:
          assert fired[a - 1]; miners.append((a, b))
        elif typ == EXTRACTION:
          want[a - 1] += 1.0 if b == 1 else 8.0; extracted.append((a, b))
        else:
          assert typ == PAIR; pairs.append((a, b >> 2, b & 3))
      assert np.array_equal(o.rewards(), want)
      # iron: mined and extracted by the same hit; gold: two different miners, both ways
      assert sorted(m for m in miners if m[1] == 1) == sorted(e for e in extracted if e[1] == 1)
      gold = sorted(a for a, b in extracted if b == 2)
      assert len(gold) % 2 == 0 and len(pairs) == len(gold)
      for a, other, typ in pairs:
        assert typ == 2 and a != other and (other, a, 2) in pairs
      gold_extractions += len(gold) // 2
      # an ore that appeared this frame appeared where no avatar stood before the moves
      new = [(c // 27, c % 27) for c in cells
             if prev_grid[ore_layer].reshape(-1)[c] == s_wait and grid[ore_layer].reshape(-1)[c] != s_wait]
      prev_pos = {(int(y), int(x)) for x, y in prev_avat[:, :2]} if step else set()
      assert not (set(new) & prev_pos)
      assert int(glob[3]) == int((grid[ore_layer].reshape(-1)[cells] != s_wait).sum())
      prev_grid, prev_avat = grid, avat
    o.close()
  assert seen[MINING] > 200 and seen[EXTRACTION] > 100 and gold_extractions >= 3, (seen, gold_extractions)


def _scripted(pk, players=3):
  """A world whose avatars the test places: returns (oracle, tables, a free ore cell with
  free cells to its left and right and two rows below)."""
  o = oracle.Oracle(pk, 5, players); o.reset()
  return o, pack.loads(pk)


def _set_ore(o, t, cell, state):
  """(test hook) the ore of `cell` in `state`: grown by a frame of the patched pack would be
  random; the oracle's piece table is reached through place/peek helpers instead."""
  o.set_cell_state(int(t["state_layer"][state]), cell % 27, cell // 27, state)


def test_two_miners_extract_gold_and_one_alone_does_not(coop_mining_pack):
  """Scripted (components.lua:107-143): gold needs 2 miners inside its window of 3 frames.
  A alone: goldRaw -> goldPartial, reward 0 (mining pays the role 'none' nothing), and 3
  frames later goldRaw again with the miners forgotten.  A, then B inside the window: both
  get 8.0 in B's frame, two pair events, the ore is gone."""
  if not hasattr(oracle.Oracle, "set_cell_state"):
    pytest.skip("oracle without the set_cell_state test hook")
  pk = util.patch_pack(coop_mining_pack, tables={"cm_thr": [0, 0, 0]})   # nothing grows, no episode end
  o, t = _scripted(pk)
  s_wait, s_iron, s_gold, _, s_part = (int(x) for x in t["cm_states"])
  cell = 13 * 27 + 4          # row 13: "WOOWWWWOOOO...": (x=1,2 ore) -> use row 10 instead
  cell = 10 * 27 + 5          # "WOOOOOOOOOWOOOOO...": x=5 is ore, so are x=4 and x=6
  ore_layer = int(t["state_layer"][s_wait])
  state_at = lambda: int(o.dump()[0][ore_layer, cell // 27, cell % 27])
  NOOP, MINE = 0, 7
  # A stands left of the ore facing east, B right of it facing west, C far away
  assert o.place_avatar(0, 4, 10, 1) and o.place_avatar(1, 6, 10, 3) and o.place_avatar(2, 20, 20, 0)
  _set_ore(o, t, cell, s_gold)
  o.step(np.array([MINE, NOOP, NOOP], np.int32))
  assert state_at() == s_part and list(o.rewards()) == [0, 0, 0]
  assert o.events() == [(MINING, 1, 2)]
  for k in range(2):
    o.step(np.array([NOOP, NOOP, NOOP], np.int32)); assert state_at() == s_part
  o.step(np.array([NOOP, NOOP, NOOP], np.int32))
  assert state_at() == s_gold                      # the window ran out: back to raw
  assert int(o.dump()[2][6]) == 0                  # ... and the miners are forgotten
  # A mines again (its cooldown of 3 is over), B two frames later
  o.step(np.array([MINE, NOOP, NOOP], np.int32)); assert state_at() == s_part
  o.step(np.array([NOOP, NOOP, NOOP], np.int32))
  o.step(np.array([NOOP, MINE, NOOP], np.int32))
  assert list(o.rewards()) == [8.0, 8.0, 0.0] and state_at() == s_wait
  assert sorted(o.events()) == sorted([(MINING, 2, 2), (EXTRACTION, 1, 2), (EXTRACTION, 2, 2),
                                       (PAIR, 1, (2 << 2) | 2), (PAIR, 2, (1 << 2) | 2)])
  # both on one IRON ore in the same frame: each hit mines and extracts (the setState the
  # first hit queues is processed a flush later: the second still finds ironRaw)
  _set_ore(o, t, cell, s_iron)
  for _ in range(3):
    o.step(np.array([NOOP, NOOP, NOOP], np.int32))
  o.step(np.array([MINE, MINE, NOOP], np.int32))
  assert list(o.rewards()) == [1.0, 1.0, 0.0] and state_at() == s_wait
  o.close()
