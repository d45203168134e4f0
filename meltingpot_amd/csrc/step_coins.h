// step_coins.h — one environment step (or episode start) of one coins world by
// one wavefront (shape: step_clean_up.h).
//
// Substrate rules restated here (reference: configs/substrates/coins.py,
// lua/levels/coins/components.lua):
//   Coin              :52-170  an avatar entering a live coin's cell collects it:
//                              reward to itself and to the others depending on
//                              whether the coin has its own colour, the partner's
//                              tracker is told, the coin waits from the next flush
//   ChoiceCoinRegrow  :173-201 a waiting coin comes back with probability
//                              regrowRate per frame, in one of the two colours
//   PartnerTracker    :281-328 "MISMATCHED_COIN_COLLECTED_BY_PARTNER" (this frame)
//   Role              :227-278 reward multipliers (folded into the pack's rewards)
//   StochasticIntervalEpisodeEnding  component_library.lua:907-948
// There is no Zapper: avatars never leave the grid.
#ifndef MP_STEP_COINS_H_
#define MP_STEP_COINS_H_

#include "step_common.h"

namespace stepk {

constexpr int kCoinRegs = 8;   // mp_create admits at most 512 coin sites
constexpr uint32_t kCoinColourDraw = 0x10000u;   // draw index of a world's colour pair

struct CoinsSites { int coin[kCoinRegs]; };   // cell of site k * 64 + lane, or -1

__device__ inline CoinsSites load_sites(const CoinsTables& c, int lane) {
  CoinsSites s;
#pragma unroll
  for (int k = 0; k < kCoinRegs; ++k) {
    const int i = k * 64 + lane;
    s.coin[k] = i < c.n_coin ? c.coin_cells[i] : -1;
  }
  return s;
}

__device__ inline void step_world(const DevTables& t, const CoinsTables& c,
                                  const CoinsSites& sites, const World& wd, const Action& act,
                                  const StepArgs& args) {
  const int lane = wd.lane, w = wd.w;
  const StepOutputs& out = args.out;
  Scratch* sc = wd.sc;
  uint8_t* grid = wd.rec;
  WorldTail* tail = reinterpret_cast<WorldTail*>(wd.rec + t.grid_pad);
  const int P = t.P, HW = t.H * t.W, W = t.W;
  const bool is_av = lane < P;
  auto at = [&](int layer, int cell) -> uint8_t& { return grid[layer * HW + cell]; };

  const OrderStreams kOrders = {RS_SHUFFLE_MOVE, 0, 0, 0, 1};   // the updater groups shuffled per frame (A1)

  const int what = dispatch(t, tail, lane, w, args.reset_mask, args.mode, args.auto_reset, out);
  if (what == 0) return;

  Av a;
  double aux0 = 0.0;
  int step_type;
  // The two coin colours of this WORLD (coins.py:500: random.sample(COIN_PALETTES,
  // k=2) when the environment is built; player 1 and coin type A wear the first,
  // player 2 and type B the second): one per-world draw of the 20 ordered pairs
  // (the map choices' stream, an index of its own, no episode in the counter), the
  // same at every reset; kept in the tail between steps.
  int s_coin0 = c.s_coin[0], s_coin1 = c.s_coin[1];
  int alive_state = is_av ? t.alive_state[lane] : 0;
  int pair = 0;
  if (c.has_colours) {
    if (what == 1) {
      pair = (int)philox_bounded(philox4x32_10(kCoinColourDraw, RS_MAP_CHOICE, 0u, 0xffffffffu,
                                               (uint32_t)tail->seed, (uint32_t)(tail->seed >> 32)),
                                 20u);
    } else {
      pair = tail->group_change;
    }
    pair = __builtin_amdgcn_readfirstlane(pair);
    const int ca = pair >> 2, r = pair & 3, cb = r + (r >= ca ? 1 : 0);
    s_coin0 = (int)((c.colour_coin >> (8 * ca)) & 255u);
    s_coin1 = (int)((c.colour_coin >> (8 * cb)) & 255u);
    const int a0 = (int)((c.colour_alive[0] >> (8 * ca)) & 255u);
    const int a1 = (int)((c.colour_alive[1] >> (8 * cb)) & 255u);
    alive_state = lane == 0 ? a0 : lane == 1 ? a1 : 0;
  }

  if (what == 1) {
    // ---- api:start (api_factory.lua:85-102); the episode number is a word of
    // the draw counter (A10; the reference re-seeds with seed + 1, builder.py:177-181)
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
      tail->aux_count = 0;  // every coin starts in coinWait
      tail->group_change = pair;   // the world's colour pair (0 without colour tables)
      tail->ctr[2]++;
    }
    if (lane < MP_MAX_PLAYERS) { tail->flag0[lane] = 0; tail->flag1[lane] = 0; }
    wsync();
    apply_map_choices(t, grid, lane, ep, k0, k1);
    spawn_avatars(t, grid, lane, ep, k0, k1, a, c.has_colours ? alive_state : -1);
    if (is_av) push_event(sc, MP_EVENT_AVATAR_STARTED, 0, 0);
    // The grid:update that ends api:start (api_factory.lua:101) runs the updaters
    // once: ChoiceCoinRegrow (components.lua:190-201) draws for every waiting coin
    // — all of them — in step 0 of the episode, so a world can start with coins.
    int live = 0;
#pragma unroll
    for (int r = 0; r < kCoinRegs; ++r) {
      const int i = r * 64 + lane;
      bool grown = false;
      if (sites.coin[r] >= 0 && at(c.wait_layer, sites.coin[r]) == c.s_wait &&
          philox_u53(philox4x32_10((uint32_t)i, RS_REGROW, 0u, ep, k0, k1)) < c.thr_regrow) {
        const uint32_t k = philox_bounded(philox4x32_10((uint32_t)i, RS_COIN_CHOICE, 0u, ep, k0, k1), 2u);
        at(c.wait_layer, sites.coin[r]) = 0;
        at(c.coin_layer, sites.coin[r]) = (uint8_t)(k == 0 ? s_coin0 : s_coin1);
        grown = true;
      }
      if (r * 64 < c.n_coin) live += __popcll(__ballot(grown));
    }
    if (lane == 0) tail->aux_count = live;
    step_type = 0;
  } else {
    // ================= api:advance =================
    const uint32_t k0 = (uint32_t)tail->seed, k1 = (uint32_t)(tail->seed >> 32);
    const uint32_t ep = tail->episode - 1;
    const int step = tail->step + 1, frame = tail->frame;
    load_avatars(tail, lane, a);
    wsync();
    auto draw = [&](int stream, uint32_t index) {
      return philox4x32_10(index, (uint32_t)stream, (uint32_t)step, ep, k0, k1);
    };
    // ---- updaters (pre-flush state)
    int orders[4];
    step_orders(tail, lane, P, kOrders, (uint32_t)step, ep, k0, k1, orders);
    const int order_move = orders[0];
    int cont = tail->cont;  // StochasticIntervalEpisodeEnding: _t == step + 1
    if (frame >= c.ee_min_frames && (step + 1) % c.ee_interval == 0)
      if (philox_u53(draw(RS_EPISODE_END, 0)) < c.thr_ee) cont = 0;
    // ChoiceCoinRegrow: a draw per waiting coin (A12), then random:choice of the
    // colour; the setState is queued behind the moves of this flush
    uint32_t regrow = 0;          // 2 bits per coin of this lane: 0 none, 1 + colour
#pragma unroll
    for (int r = 0; r < kCoinRegs; ++r) {
      const int i = r * 64 + lane;
      if (sites.coin[r] < 0) continue;
      if (at(c.wait_layer, sites.coin[r]) != c.s_wait) continue;
      if (philox_u53(draw(RS_REGROW, (uint32_t)i)) >= c.thr_regrow) continue;
      regrow |= (1u + philox_bounded(draw(RS_COIN_CHOICE, (uint32_t)i), 2u)) << (2 * r);
    }

    // ---- flush 1: the moves, in visiting order
    const bool wants = resolve_moves(t, wd, a, act.move, act.turn, order_move, alive_state);
    // Coin:onEnter on the cell the avatar is in now (A3b: a blocked move
    // re-enters its own cell)
    int got = -1, got_cell = -1;   // colour of the coin this avatar collected
    if (wants) {
      const int s = at(c.coin_layer, a.y * W + a.x);
      if (s == s_coin0 || s == s_coin1) { got = s == s_coin1; got_cell = a.y * W + a.x; }
    }
    for (int p = 0; p < P; ++p) {   // rewards: collector and everyone else
      const int gp = rdlane(got, p);
      if (gp < 0) continue;
      const int ptype = p == 0 ? c.player_type[0] : c.player_type[1];
      const bool match = gp == ptype;
      const double* rw = p == 0 ? c.rew[0] : c.rew[1];
      if (lane == p) {
        a.reward += match ? rw[0] : rw[1];
        push_event(sc, MP_EVENT_COIN_CONSUMED, p + 1, (ptype << 1) | gp);
      } else if (is_av) {
        a.reward += match ? rw[2] : rw[3];          // Coin:rewardOthers
        if (!match && lane == (p == 0 ? 1 : 0)) aux0 = 1.0;  // PartnerTracker:reportMismatch
      }
    }
    wsync();
    // the regrown coins appear (the last events of flush 1)
#pragma unroll
    for (int r = 0; r < kCoinRegs; ++r) {
      const uint32_t g = (regrow >> (2 * r)) & 3u;
      if (!g) continue;
      const int cell = sites.coin[r];
      at(c.wait_layer, cell) = 0;
      at(c.coin_layer, cell) = (uint8_t)(g == 1u ? s_coin0 : s_coin1);
    }
    wsync();
    // ---- flush 2: collected coins go to coinWait
    if (got_cell >= 0) {
      at(c.coin_layer, got_cell) = 0;
      at(c.wait_layer, got_cell) = (uint8_t)c.s_wait;
    }
    wsync();
    int live = 0;
#pragma unroll
    for (int r = 0; r < kCoinRegs; ++r) {
      if (r * 64 >= c.n_coin) break;
      live += __popcll(__ballot(sites.coin[r] >= 0 &&
                                at(c.coin_layer, sites.coin[r] >= 0 ? sites.coin[r] : 0) != 0));
    }
    const unsigned long long badb = __ballot(act.bad != 0);
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
  // "N.MISMATCHED_COIN_COLLECTED_BY_PARTNER" is the substrate metric; no Zapper:
  // READY_TO_SHOOT stays 1 (timer 0, cooldown 1)
  finish(t, wd, tail, a, aux0, 1, step_type, out, kOrders);
}

}  // namespace stepk

#endif  // MP_STEP_COINS_H_
