"""gift_refinements (lua/levels/gift_refinements/components.lua,
configs/substrates/gift_refinements.py): the committed pack against the reference config, and
the oracle's restatement of the rules — an independent Python model of the inventories run
next to oracle rollouts with plentiful tokens (consume, pick, refine-and-gift in the order
the frame delivers them), and a scripted gift with hand-computed expectations."""
import os

import numpy as np
import pytest

import util
from meltingpot_amd import engine, lower, pack, refshim
from oracle import oracle

HAVE_REFERENCE = os.path.isdir("/root/reference/meltingpot")
GIFT = 16
GIFT_HEAVY = [1, 4, 1, 1, 1, 2, 2, 5, 2]   # weights over the ACTION_SET: gifts often, consumes rarely


@pytest.fixture(scope="module")
def gift_pack() -> bytes:
  return engine.load_pack("gift_refinements")


def rich(pack_bytes, rate=0.02):
  """The pack with tokens growing 100 x faster (the stock rate, 2e-4 per site and frame,
  leaves random play almost nothing to gift)."""
  t = pack.loads(pack_bytes)
  thr = t["gr_thr"].copy()
  thr[0] = lower.prob_threshold(rate)
  return util.patch_pack(pack_bytes, tables={"gr_thr": thr})


def decode_gift(a, b):
  """event row -> (gifter, source type, recipient, the count the recipient now holds)."""
  return (a & 15) - 1, (a >> 4) - 1, (b & 15) - 1, b >> 4


@pytest.mark.skipif(not HAVE_REFERENCE, reason="reference tree not present (GPU box)")
def test_committed_pack_is_what_the_reference_config_lowers_to(gift_pack):
  settings, mod, config = refshim.build_settings("gift_refinements", ("default",) * 8)
  assert pack.dumps(lower.lower("gift_refinements", settings, mod.ACTION_SET, default_players=6)) == (
      gift_pack), "run tools/make_packs.py"
  t = pack.loads(gift_pack)
  hdr = t["hdr"]
  assert (hdr[lower.HDR_H], hdr[lower.HDR_W]) == (27, 27)                    # gift_refinements.py:69-97
  assert hdr[lower.HDR_MAXFRAMES] == 5000 and hdr[lower.HDR_DEFAULT_P] == 6  # :497, :475
  assert len(t["token_cells"]) == mod.ASCII_MAP.count("T")
  names = bytes(t["state_names"]).split(b"\0")
  assert [names[s] for s in t["gr_states"]] == [b"token.tokenWait", b"token.token"]
  # cooldown 3, beam 5 x 0, hit 0, episode ending, capacity 15, 3 token types, multiplier 5,
  # no consumption cooldown (:338-358, :125-134)
  assert list(t["gr_i32"]) == [3, 5, 0, 0, 1000, 100, mod.MAX_TOKENS_PER_TYPE, mod.NUM_TOKEN_TYPES, 5, 0]
  f = t["gr_f64"]
  assert list(f[:16]) == [0.0] * 16            # role "none": roleRewardForGifting 0.0
  assert list(f[-3:]) == [0.0, 0.0002, 0.2]    # rewardForPicking, regrowRate, episode end
  assert list(t["gr_thr"]) == [lower.prob_threshold(p) for p in (0.0002, 0.2)]
  assert bytes(t["hit_names"]) == b"gift\0"
  assert bytes(t["action_names"]) == b"move\0turn\0refineAndGift\0consumeTokens\0"
  s2, _, _ = refshim.build_settings("gift_refinements", ("target",) * 8)
  assert pack.dumps(lower.lower("gift_refinements", s2, mod.ACTION_SET, default_players=6)) == gift_pack


def test_rules_hold_over_rollouts_with_plentiful_tokens(gift_pack):
  """Every frame of six rollouts: consumption pays exactly what the avatar held and empties
  it first (Inventory:update runs in BaseSimulation:update); a `gift` event names a gifter that
  fired this frame, another avatar as recipient and the count the recipient then holds; tokens
  are conserved type by type (type 1 enters by picking only — at most one per avatar that moved
  onto a token cell now empty —, type k + 1 only as 5 per refined type-k gift, capacity 15);
  READY_TO_SHOOT follows the cooldown of 3; tokens only grow where no avatar stands."""
  pk = rich(gift_pack)
  t = pack.loads(pk)
  s_wait, s_live = (int(x) for x in t["gr_states"])
  layer = int(t["state_layer"][s_live])
  rng = np.random.default_rng(4)
  gifts = refined = picked = consumed = capped = 0
  P = 6
  for seed in range(6):
    o = oracle.Oracle(pk, util.world_seed(seed), P); o.reset()
    inv = np.zeros((P, 3), int)
    timer = np.zeros(P, int)
    prev_grid = o.dump()[0]
    assert np.array_equal(o.inventories()[0], inv)
    for step in range(500):
      acts = rng.choice(9, size=P, p=np.array(GIFT_HEAVY) / sum(GIFT_HEAVY)).astype(np.int32)
      reward = np.zeros(P)
      for p in range(P):                          # Inventory:update, then GiftBeam:update
        if acts[p] == 8:
          reward[p] += inv[p].sum(); consumed += inv[p].sum(); inv[p] = 0
        timer[p] = max(timer[p] - 1, 0)
      fired = {p for p in range(P) if acts[p] == 7 and timer[p] == 0}
      for p in fired:
        timer[p] = 3
      o.step(acts)
      grid, avat, glob = o.dump()
      got = np.array([[(avat[p, 7] >> (4 * k)) & 15 for k in range(3)] for p in range(P)])
      assert np.array_equal(o.inventories()[0], got.astype(float))
      assert np.array_equal(o.rewards(), reward)      # stock config: every other reward is 0.0
      assert np.array_equal(avat[:, 4], timer)
      assert np.allclose(o.ready_to_shoot(), 1.0 - timer / 3.0)
      ev = [decode_gift(a, b) for ty, a, b in o.events() if ty == GIFT]
      by_src = [0, 0, 0]
      for g, src, r, cnt in ev:
        assert g in fired and r != g and 0 <= r < P and 0 <= src < 3 and 1 <= cnt <= 15
        by_src[src] += 1
      assert len({g for g, *_ in ev}) == len(ev)     # one beam, one recipient per gifter
      gifts += len(ev); refined += by_src[0] + by_src[1]
      # conservation (exact while nothing sits at the capacity of 15, where additions are cut)
      if got.max() == 15:
        capped += 1
        inv, prev_grid = got, grid
        continue
      could_pick = sum(1 for p in range(P) if acts[p] in (1, 2, 3, 4)
                       and grid[layer, avat[p, 1], avat[p, 0]] == s_wait)
      picks = got[:, 0].sum() - inv[:, 0].sum() + by_src[0]
      assert 0 <= picks <= could_pick
      picked += picks
      assert got[:, 1].sum() == inv[:, 1].sum() + 5 * by_src[0] - by_src[1]
      assert got[:, 2].sum() == inv[:, 2].sum() + 5 * by_src[1]      # (a type-3 gift moves it)
      grown = (grid[layer] == s_live) & (prev_grid[layer] == s_wait)
      for p in range(P):
        assert not grown[avat[p, 1], avat[p, 0]]
      inv, prev_grid = got, grid
  assert gifts > 40 and refined > 20 and picked > 100 and consumed > 50 and capped < 300, (
      gifts, refined, picked, consumed, capped)


def test_a_scripted_refinement_chain(gift_pack):
  """Player 0 picks a raw token and faces player 1 three cells away: the gift turns it into
  five tokens of the next type (event: source type 1, the recipient now holds 5); player 1
  gifts back — its highest type, 2, becomes five of type 3 — and a third gift of a type-3 token
  passes it on unrefined (count + 1, no multiplier); consumption pays the whole inventory."""
  t = pack.loads(gift_pack)
  o = oracle.Oracle(gift_pack, util.world_seed(1), 2); o.reset()
  s_live = int(t["gr_states"][1])
  W = int(t["hdr"][lower.HDR_W])
  # an open row of the map: y = 1, x = 1 .. 25 are token cells (gift_refinements.py:71)
  assert o.place_avatar(0, 3, 1, 1)      # facing E
  assert o.place_avatar(1, 7, 1, 3)      # facing W
  cell = 1 * W + 4
  assert cell in t["token_cells"]
  assert o.set_cell_state(int(t["state_layer"][s_live]), 4, 1, s_live)
  noop = np.zeros(2, np.int32)
  o.step(np.array([1, 0], np.int32))                 # 0 steps E onto the live token: picked
  inv = o.inventories()[0]
  assert inv.tolist() == [[1, 0, 0], [0, 0, 0]]
  o.step(np.array([7, 0], np.int32))                 # 0 gifts: 1 x type 1 -> 5 x type 2
  assert o.inventories()[0].tolist() == [[0, 0, 0], [0, 5, 0]]
  ev = [e for e in o.events() if e[0] == GIFT]
  assert [decode_gift(a, b) for _, a, b in ev] == [(0, 0, 1, 5)]
  assert o.ready_to_shoot().tolist() == [0.0, 1.0]
  o.step(np.array([0, 7], np.int32))                 # 1 gifts back: 1 x type 2 -> 5 x type 3
  assert o.inventories()[0].tolist() == [[0, 0, 5], [0, 4, 0]]
  assert [decode_gift(a, b) for _, a, b in o.events() if _ == GIFT] == [(1, 1, 0, 5)]
  o.step(noop); o.step(noop)
  o.step(np.array([7, 0], np.int32))                 # 0 gifts a type-3 token: passed on as it is
  assert o.inventories()[0].tolist() == [[0, 0, 4], [0, 4, 1]]
  assert [decode_gift(a, b) for _, a, b in o.events() if _ == GIFT] == [(0, 2, 1, 1)]
  o.step(np.array([8, 8], np.int32))                 # both consume
  assert o.rewards().tolist() == [4.0, 5.0]
  assert o.inventories()[0].tolist() == [[0, 0, 0], [0, 0, 0]]
