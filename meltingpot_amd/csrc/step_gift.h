// step_gift.h — one environment step (or episode start) of one gift_refinements world by
// one wavefront (shape: step_clean_up.h).
//
// Substrate rules restated here (reference: configs/substrates/gift_refinements.py,
// lua/levels/gift_refinements/components.lua):
//   FixedRateRegrow :29-55    a component update(), NOT an engine-side updater: in
//                             BaseSimulation:update every token in tokenWait draws; a hit
//                             queues setState(token) unless an avatar stands on it — AHEAD of
//                             every updater's events of the frame
//   Pickable        :57-90    an avatar entering a live token's cell takes it: one token of
//                             type 1 into its inventory, the token to tokenWait (next flush)
//   Inventory       :239-353  per-type counts up to capacityPerType; update(): the consume
//                             action pays the whole inventory and empties it — in
//                             BaseSimulation:update, i.e. BEFORE this frame's picks and gifts
//   GiftBeam        :92-237   cooldown timer run down in update(); the updater (priority
//                             140, behind the moves of 150) fires the beam; a hit avatar stops
//                             it and receives: the gifter loses one token of its HIGHEST type
//                             k, the recipient gains giftMultiplier tokens of type k + 1 (or
//                             that one token, if k is the most refined).  addTokens RETURNS
//                             THE NEW COUNT (:307-318) and that is what the event reports.
//   StochasticIntervalEpisodeEnding  component_library.lua:907-948
// Gifts are delivered beam by beam in the gift updater's visiting order: a gift changes what
// the next gifter's highest type is and how much room the next recipient has.
#ifndef MP_STEP_GIFT_H_
#define MP_STEP_GIFT_H_

#include "step_common.h"

namespace stepk {

constexpr int kTokenRegs = 10;   // mp_create admits at most 640 token sites

struct GiftSites { int token[kTokenRegs]; };

__device__ inline GiftSites load_sites(const GiftTables& c, int lane) {
  GiftSites s;
#pragma unroll
  for (int k = 0; k < kTokenRegs; ++k) {
    const int i = k * 64 + lane;
    s.token[k] = i < c.n_token ? c.token_cells[i] : -1;
  }
  return s;
}

__device__ inline void step_world(const DevTables& t, const GiftTables& c,
                                  const GiftSites& sites, const World& wd, const Action& act,
                                  const StepArgs& args) {
  const int lane = wd.lane, w = wd.w;
  const StepOutputs& out = args.out;
  Scratch* sc = wd.sc;
  uint8_t* grid = wd.rec;
  WorldTail* tail = reinterpret_cast<WorldTail*>(wd.rec + t.grid_pad);
  const int P = t.P, HW = t.H * t.W, W = t.W;
  const bool is_av = lane < P;
  auto at = [&](int layer, int cell) -> uint8_t& { return grid[layer * HW + cell]; };

  const OrderStreams kOrders = {RS_SHUFFLE_MOVE, RS_SHUFFLE_ZAP, 0, 0, 2};   // the updater groups shuffled per frame (A1)

  const int what = dispatch(t, tail, lane, w, args.reset_mask, args.mode, args.auto_reset, out);
  if (what == 0) return;

  Av a;
  int step_type;
  int inv0 = 0, inv1 = 0, inv2 = 0;   // Inventory.inventory of avatar `lane`
  const int alive_state = is_av ? t.alive_state[lane] : 0;
  const int s_wait = c.s_wait, s_live = c.s_live;

  if (what == 1) {
    // ---- api:start (api_factory.lua:85-102); the episode number is a word of the draw
    // counter (A10; the reference re-seeds with seed + 1, builder.py:177-181)
    const uint32_t k0 = (uint32_t)tail->seed, k1 = (uint32_t)(tail->seed >> 32);
    const uint32_t ep = tail->episode;
    wsync();
    const int gvec = t.grid_pad >> 4;
    for (int i = lane; i < gvec; i += 64)
      reinterpret_cast<uint4*>(grid)[i] = reinterpret_cast<const uint4*>(t.init_grid)[i];
    if (lane == 0) {
      tail->episode = ep + 1;
      tail->step = 0; tail->frame = 1; tail->done = 0; tail->cont = 1;
      tail->started = 1;
      tail->aux_count = 0;     // (every token starts in tokenWait)
      tail->group_change = 0;
      tail->ctr[2]++;
    }
    wsync();
    apply_map_choices(t, grid, lane, ep, k0, k1);
    spawn_avatars(t, grid, lane, ep, k0, k1, a);
    if (lane < P) push_event(sc, MP_EVENT_AVATAR_STARTED, 0, 0);
    // (Inventory:reset / :start, GiftBeam:start: empty, both timers 0 — `a` starts cleared.
    // The grid:update of api:start runs no simulation:update: no token grows at frame 0)
    step_type = 0;
  } else {
    // ================= api:advance =================
    const uint32_t k0 = (uint32_t)tail->seed, k1 = (uint32_t)(tail->seed >> 32);
    const uint32_t ep = tail->episode - 1;
    const int step = tail->step + 1, frame = tail->frame;
    load_avatars(tail, lane, a);
    if (lane < MP_MAX_PLAYERS) { inv0 = tail->flag0[lane]; inv1 = tail->flag1[lane]; inv2 = tail->level[lane]; }
    wsync();
    const int a_move = act.move, a_turn = act.turn, a_gift = act.fire0, a_consume = act.fire1,
              bad = act.bad;

    // ---- BaseSimulation:update, objects in creation order: the avatars ...
    if (is_av) {
      // Inventory:update (components.lua:328-350); ctimer > 0 <=> _consumeCooldownTimer > 0
      if (a_consume == 1 && a.ctimer == 0) {
        a.reward += (double)(inv0 + inv1 + inv2);
        inv0 = inv1 = inv2 = 0;
        a.ctimer = c.consume_cooldown;
      }
      if (a.ctimer > 0) a.ctimer--;
      // GiftBeam:update (components.lua:222-226)
      if (a.ztimer > 0) a.ztimer--;
    }
    // ... then the tokens: FixedRateRegrow:update — one draw per WAITING token, the avatars
    // looked for where they stand NOW (before this frame's moves); the setState is queued
    // ahead of the moves: an avatar that steps onto the cell this frame finds the token live
#pragma unroll
    for (int r = 0; r < kTokenRegs; ++r) {
      if (r * 64 >= c.n_token) break;
      const int cell = sites.token[r];
      if (cell < 0) continue;
      if (at(c.token_layer, cell) != s_wait) continue;
      if (philox_u53(philox4x32_10((uint32_t)(r * 64 + lane), RS_REGROW, (uint32_t)step, ep, k0, k1)) <
              c.thr[0] &&
          at(t.avatar_layer, cell) == 0)
        at(c.token_layer, cell) = (uint8_t)s_live;
    }
    // ---- updaters (pre-flush state)
    int orders[4];
    step_orders(tail, lane, P, kOrders, (uint32_t)step, ep, k0, k1, orders);
    const int order_move = orders[0], order_gift = orders[1];
    // 140 GiftBeam gift (components.lua:186-211)
    bool fire = false;
    if (is_av && a.alive && a_gift == 1 && a.ztimer == 0) { a.ztimer = c.cooldown; fire = true; }
    int cont = tail->cont;  // StochasticIntervalEpisodeEnding: _t == step + 1
    if (frame >= c.ee_min_frames && (step + 1) % c.ee_interval == 0)
      if (philox_u53(philox4x32_10(0u, RS_EPISODE_END, (uint32_t)step, ep, k0, k1)) < c.thr[1]) cont = 0;
    // beam sprites of the previous frame disappear (grid:update start)
    clear_bytes(grid, c.beam_layer * HW, HW, lane);
    wsync();

    // ---- flush 1: the moves, in visiting order ...
    const bool wants = resolve_moves(t, wd, a, a_move, a_turn, order_move, alive_state);
    // Pickable:onEnter on the destination — or, for a blocked move, on the cell the avatar
    // stays in (A3b); the token goes to tokenWait in the next flush
    int picked_cell = -1;
    if (wants && at(c.token_layer, a.y * W + a.x) == s_live) {
      a.reward += c.pick_reward;
      inv0 = min(inv0 + 1, c.capacity);      // Inventory:addTokens(1, 1)
      picked_cell = a.y * W + a.x;
    }
    wsync();
    // ... then the beams.  No gift changes where anybody stands: the footprints are
    // evaluated together; what a hit DOES follows the queue, beam by beam in the gift
    // updater's visiting order, against the inventories as the earlier gifts left them.
    fire_beams(t, wd, tail, a, fire, beam_lane(c.shape, lane), c.hit, false,
               c.beam_layer, c.s_beam, false,
               // GiftBeam:onHit (components.lua:135-184): a hit avatar stops the beam
               [&](int s, int) { const int pl = (int)(wd.sinfo[s] >> 24); return pl ? 1 | (pl << 8) : 0; },
               [](int, int, int, bool, int, bool) {});
    {
      const int firing = (int)fire;
      const int nc = c.shape.n;
      if (__ballot(fire) != 0ull) {
        for (int r = 0; r < P; ++r) {
          const int g = rdlane(order_gift, r);
          if (rdlane(firing, g) == 0) continue;
          int victim = -1;
          for (int q = 0; q < nc; ++q) { const int v = sc->victim[g][q]; if (v >= 0) victim = v; }
          if (victim < 0) continue;
          // hitterAvatar:addReward(roleRewardForGifting[role]) — every hit
          if (lane == g) a.reward += c.reward[2 * g];
          // Inventory:getHighestTypeAvailable of the gifter
          const int g0 = rdlane(inv0, g), g1 = rdlane(inv1, g), g2 = rdlane(inv2, g);
          int src = -1;
          if (g0 > 0) src = 0;
          if (g1 > 0 && c.ntypes > 1) src = 1;
          if (g2 > 0 && c.ntypes > 2) src = 2;
          if (src < 0) continue;
          int dst = src + 1, amount = c.multiplier;
          if (dst >= c.ntypes) { dst = c.ntypes - 1; amount = 1; }           // the most refined: passed on as it is
          else if (lane == g) a.reward += c.reward[2 * g + 1];              // amount * successfulGiftReward
          if (lane == g) { if (src == 0) inv0--; else if (src == 1) inv1--; else inv2--; }
          const int have = dst == 0 ? rdlane(inv0, victim) : dst == 1 ? rdlane(inv1, victim) : rdlane(inv2, victim);
          const int now = min(have + amount, c.capacity);                    // addTokens: the NEW count
          if (lane == victim) { if (dst == 0) inv0 = now; else if (dst == 1) inv1 = now; else inv2 = now; }
          if (lane == g) push_event(sc, MP_EVENT_GIFT, (g + 1) | ((src + 1) << 4), (victim + 1) | (now << 4));
        }
      }
    }

    // ---- flush 2: the picked tokens wait
    if (picked_cell >= 0) at(c.token_layer, picked_cell) = (uint8_t)s_wait;
    wsync();
    int live = 0;
#pragma unroll
    for (int r = 0; r < kTokenRegs; ++r) {
      if (r * 64 >= c.n_token) break;
      const int cell = sites.token[r];
      live += __popcll(__ballot(cell >= 0 && at(c.token_layer, cell >= 0 ? cell : 0) == s_live));
    }
    const unsigned long long badb = __ballot(bad != 0);
    const int done = !(cont && step < t.max_frames);
    if (lane == 0) {
      tail->step = step;
      tail->frame = frame + 1;
      tail->cont = cont;
      tail->done = done;
      tail->aux_count = live;
      tail->ctr[0]++; tail->ctr[1] += (uint32_t)P; tail->ctr[7] += __popcll(badb);
    }
    step_type = done ? 2 : 1;
  }
  if (lane < MP_MAX_PLAYERS) {
    tail->flag0[lane] = (uint8_t)inv0; tail->flag1[lane] = (uint8_t)inv1; tail->level[lane] = (uint8_t)inv2;
  }
  // "N.INVENTORY" (AvatarMetricReporter on Inventory.inventory, gift_refinements.py:367-380)
  if (is_av && out.inventory) {
    const size_t o = ((size_t)w * P + lane) * c.ntypes;
    if (c.ntypes > 0) out.inventory[o] = (double)inv0;
    if (c.ntypes > 1) out.inventory[o + 1] = (double)inv1;
    if (c.ntypes > 2) out.inventory[o + 2] = (double)inv2;
  }
  // READY_TO_SHOOT reads the GiftBeam (ReadyToShootObservation.zapperComponent)
  finish(t, wd, tail, a, 0.0, c.cooldown, step_type, out, kOrders);
}

}  // namespace stepk

#endif  // MP_STEP_GIFT_H_
