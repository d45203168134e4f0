// step_commons.h — one environment step (or episode start) of one
// commons_harvest world by one wavefront (shape: step_clean_up.h).
//
// Substrate rules restated here (reference: configs/substrates/
// commons_harvest__open.py, lua/levels/commons_harvest/components.lua):
//   DensityRegrow  :71-240  an eaten apple waits in state appleWait_k, k = live
//                           apples within the L2 disc of radius 2; an
//                           engine-side updater per k re-grows it with
//                           probability p[min(k, 3)]; k == 0 dessicates the grass
//   Edible         component_library.lua:953-1004
//   StochasticIntervalEpisodeEnding  component_library.lua:907-948
// The reference keeps k incrementally (Neighborhoods.pieceToNumNeighbors,
// updated by _beginLive/_endLive callbacks); the invariant those callbacks
// maintain — for every waiting apple, k = number of live apples in its disc —
// is evaluated here directly from the grid (a 12-cell stencil per apple, one
// apple per lane), so the kernel carries no per-apple counter.
#ifndef MP_STEP_COMMONS_H_
#define MP_STEP_COMMONS_H_

#include "step_common.h"

#ifdef MP_STEP_TIMING
#include <stdio.h>
#define CTSTAMP(i) cts_[i] = __builtin_readcyclecounter()
#else
#define CTSTAMP(i)
#endif

namespace stepk {

constexpr int kAppleRegs = 4;   // mp_create admits at most 256 apple sites

// This lane's apple sites (site k * 64 + lane; -1 beyond the list) and, spread
// over the lanes, the small tables the rules index per lane: lane k holds the
// appleWait_k state id and its regrowth threshold, lane d the d-th offset of the
// disc (a per-lane index into a kernel argument or a global table would be a
// memory round trip each time; a lane exchange is not).
struct CommonsSites {
  int apple[kAppleRegs];
  int wait_state;             // lane k: c.s_wait_k[k]
  uint32_t thr_lo, thr_hi;    // lane k: c.thr[k]
  int disc_dx, disc_dy;       // lane d: c.disc[d]
};

__device__ inline CommonsSites load_sites(const CommonsTables& c, int lane) {
  CommonsSites s;
#pragma unroll
  for (int k = 0; k < kAppleRegs; ++k) {
    const int i = k * 64 + lane;
    s.apple[k] = i < c.n_apple ? c.apple_cells[i] : -1;
  }
  s.wait_state = 0;
  for (int k = 0; k < c.nk; ++k) if (lane == k) s.wait_state = c.s_wait_k[k];
  const uint64_t thr = lane <= c.nk ? c.thr[lane] : 0ull;
  s.thr_lo = (uint32_t)thr; s.thr_hi = (uint32_t)(thr >> 32);
  s.disc_dx = lane < c.ndisc ? c.disc[2 * lane] : 0;
  s.disc_dy = lane < c.ndisc ? c.disc[2 * lane + 1] : 0;
  return s;
}

__device__ inline void step_world(const DevTables& t, const CommonsTables& c,
                                  const CommonsSites& sites, const World& wd, const Action& act,
                                  const StepArgs& args) {
  const int lane = wd.lane, w = wd.w;
  const StepOutputs& out = args.out;
  Scratch* sc = wd.sc;
  uint8_t* grid = wd.rec;
  WorldTail* tail = reinterpret_cast<WorldTail*>(wd.rec + t.grid_pad);
  const int P = t.P, HW = t.H * t.W, W = t.W, H = t.H;
  const bool is_av = lane < P;
  auto at = [&](int layer, int cell) -> uint8_t& { return grid[layer * HW + cell]; };

#ifdef MP_STEP_TIMING
  unsigned long long cts_[8] = {0};
#endif
  CTSTAMP(0);
  const OrderStreams kOrders = {RS_SHUFFLE_MOVE, RS_SHUFFLE_ZAP, RS_SHUFFLE_RESPAWN, 0, 3};   // the updater groups shuffled per frame (A1)
  const int what = dispatch(t, tail, lane, w, args.reset_mask, args.mode, args.auto_reset, out);
  if (what == 0) return;

  Av a;
  int step_type;
  const int alive_state = is_av ? t.alive_state[lane] : 0;
  double* zmat = out.zap_matrix ? out.zap_matrix + (size_t)w * P * P : nullptr;
  if (zmat) for (int i = lane; i < P * P; i += 64) zmat[i] = 0.0;

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
      tail->aux_count = c.n_apple;  // every apple starts live
      tail->group_change = 0;
      tail->ctr[2]++;
    }
    if (lane < MP_MAX_PLAYERS) { tail->flag0[lane] = 0; tail->flag1[lane] = 0; }
    wsync();
    apply_map_choices(t, grid, lane, ep, k0, k1);
    spawn_avatars(t, grid, lane, ep, k0, k1, a);
    if (lane < P) push_event(sc, MP_EVENT_AVATAR_STARTED, 0, 0);
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
    const int a_move = act.move, a_turn = act.turn, a_zap = act.fire0, bad = act.bad;

    // ---- per waiting apple (one lane each, up to 4 rounds):
    //  * DensityRegrow sprout updater (priority 10): decided on the state the
    //    piece has NOW (set by the previous frame), A12: one draw per piece;
    //  * DensityRegrow:update -> _updateWaitState (components.lua:161-193):
    //    k = live apples in the disc, as of the end of the previous frame.
    uint32_t sprout_bits = 0, wait_bits = 0;
    int new_k[kAppleRegs] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < kAppleRegs; ++r) {
      if (r * 64 >= c.n_apple) break;
      const int i = r * 64 + lane;
      const bool valid = sites.apple[r] >= 0;
      const int cell = valid ? sites.apple[r] : 0;
      const int ws = at(c.wait_layer, cell);
      const bool waiting = valid && ws != 0 && at(c.live_layer, cell) != c.s_apple;
      int old_k = -1;
      for (int k = 0; k < c.nk; ++k) if (ws == c.s_wait_k[k]) old_k = k;
      // (lane exchanges stay outside divergent code: every lane takes part)
      const int ok = old_k >= 0 ? old_k : 0;
      const uint64_t thr = ((uint64_t)(uint32_t)__shfl((int)sites.thr_hi, ok) << 32) |
                           (uint32_t)__shfl((int)sites.thr_lo, ok);
      if (!waiting) continue;
      wait_bits |= 1u << r;
      if (old_k >= 0 && philox_u53(draw(RS_REGROW, (uint32_t)i)) < thr)
        sprout_bits |= 1u << r;
      const int x0 = cell % W, y0 = cell / W;
      int n = 0;
      for (int d = 0; d < c.ndisc; ++d) {   // predicated: no branch between the LDS reads
        int x = x0 + rdlane(sites.disc_dx, d), y = y0 + rdlane(sites.disc_dy, d);
        bool inb = true;
        if (t.topology == 1) { x = ((x % W) + W) % W; y = ((y % H) + H) % H; }
        else inb = x >= 0 && x < W && y >= 0 && y < H;
        n += (inb && at(c.live_layer, inb ? y * W + x : cell) == c.s_apple) ? 1 : 0;
      }
      new_k[r] = n;
    }
    CTSTAMP(1);
    // beam sprites of the previous frame disappear (grid:update start)
    clear_bytes(grid, c.zap.layer * HW, HW, lane);
    wsync();
    // first events of the flush: setState(appleWait_k) and the grass under it
#pragma unroll
    for (int r = 0; r < kAppleRegs; ++r) {
      if (r * 64 >= c.n_apple) break;
      const int ns = __shfl(sites.wait_state, new_k[r]);   // c.s_wait_k[new_k]
      if (!((wait_bits >> r) & 1u)) continue;
      const int cell = sites.apple[r];
      at(c.wait_layer, cell) = (uint8_t)ns;
      const int g = at(c.grass_layer, cell);
      if (g == c.s_grass || g == c.s_dess)
        at(c.grass_layer, cell) = (uint8_t)(new_k[r] == 0 ? c.s_dess : c.s_grass);
    }

    // ---- updaters (pre-flush state)
    int orders[4];
    step_orders(tail, lane, P, kOrders, (uint32_t)step, ep, k0, k1, orders);
    const int order_move = orders[0], order_zap = orders[1], order_resp = orders[2];
    bool fire_zap = false, want_respawn = false;
    if (is_av) {
      if (a.alive && c.zap.cooldown >= 0) {  // Zapper zap (avatar_library.lua:613-636)
        if (a.ztimer > 0) a.ztimer--;
        else if (a_zap == 1) { a.ztimer = c.zap.cooldown; fire_zap = true; }
      }
      want_respawn = !a.alive && (frame - a.achange) >= c.zap.respawn_frames;
    }
    int cont = tail->cont;  // StochasticIntervalEpisodeEnding: _t == step + 1
    if (frame >= c.ee_min_frames && (step + 1) % c.ee_interval == 0)
      if (philox_u53(draw(RS_EPISODE_END, 0)) < c.thr[c.nk]) cont = 0;

    CTSTAMP(2);
    // ---- flush 1
    const bool wants = resolve_moves(t, wd, a, a_move, a_turn, order_move, alive_state);
    // Edible:onEnter (component_library.lua:990-1004); apple -> appleWait next flush
    int ate_cell = -1;
    if (wants && at(c.live_layer, a.y * W + a.x) == c.s_apple) {
      a.reward += c.eat_reward; ate_cell = a.y * W + a.x;
      push_event(sc, MP_EVENT_EDIBLE_CONSUMED, lane + 1, 0);
    }
    wsync();
    CTSTAMP(3);
    fire_beams(t, wd, tail, a, fire_zap, beam_lane(c.zap.shape, lane), c.zap.hit, true,
               c.zap.layer, c.zap.s_hit, c.zap.remove_hit != 0,
               [](int, int) { return 0; },
               [](int, int, int, bool, int, bool) {}, -1, zmat);
    zap_rewards(t, sc, lane, a, fire_zap, order_zap, c.zap.shape.n, c.zap.penalty,
                c.zap.reward);
    CTSTAMP(4);
    const int rcell = resolve_respawns(t, wd, tail, a, want_respawn, order_resp, alive_state,
                                       (uint32_t)step, frame, ep, k0, k1);
    CTSTAMP(5);
    if (rcell >= 0 && at(c.live_layer, rcell) == c.s_apple) {
      a.reward += c.eat_reward; ate_cell = rcell;
      push_event(sc, MP_EVENT_EDIBLE_CONSUMED, lane + 1, 0);
    }
    wsync();
    // sprouts: setState(apple), the last events of flush 1 (canRegrowIfOccupied)
#pragma unroll
    for (int r = 0; r < kAppleRegs; ++r) {
      if (!((sprout_bits >> r) & 1u)) continue;
      const int cell = sites.apple[r];
      if (at(c.live_layer, cell) == 0) {
        at(c.wait_layer, cell) = 0;
        at(c.live_layer, cell) = (uint8_t)c.s_apple;
      }
    }
    wsync();

    // ---- flush 2
    if (ate_cell >= 0 && at(c.wait_layer, ate_cell) == 0) {  // apple -> appleWait
      at(c.live_layer, ate_cell) = 0;
      at(c.wait_layer, ate_cell) = (uint8_t)c.s_wait;
    }
    apply_zapped(t, wd, a, rcell >= 0, frame);
    wsync();
    int live = 0;
#pragma unroll
    for (int r = 0; r < kAppleRegs; ++r) {
      if (r * 64 >= c.n_apple) break;
      live += __popcll(__ballot(sites.apple[r] >= 0 &&
                                at(c.live_layer, sites.apple[r] >= 0 ? sites.apple[r] : 0) == c.s_apple));
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
  CTSTAMP(6);
  finish(t, wd, tail, a, 0.0, c.zap.cooldown, step_type, out, kOrders);
  CTSTAMP(7);
#ifdef MP_STEP_TIMING
  if (lane == 0 && (w == 7 || w == 2000) && what == 2)
    printf("w %d: apples %llu upd %llu moves %llu zap %llu resp %llu flush2 %llu finish %llu total %llu\n",
           w, cts_[1] - cts_[0], cts_[2] - cts_[1], cts_[3] - cts_[2], cts_[4] - cts_[3],
           cts_[5] - cts_[4], cts_[6] - cts_[5], cts_[7] - cts_[6], cts_[7] - cts_[0]);
#endif
}

}  // namespace stepk

#endif  // MP_STEP_COMMONS_H_
