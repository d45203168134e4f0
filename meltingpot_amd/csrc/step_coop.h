// step_coop.h — one environment step (or episode start) of one coop_mining world by
// one wavefront (shape: step_clean_up.h).
//
// Substrate rules restated here (reference: configs/substrates/coop_mining.py,
// lua/levels/coop_mining/components.lua):
//   FixedRateRegrow :29-60    an ore in oreWait becomes ironRaw / goldRaw: one engine-side
//                             probabilistic updater per live state (priority 200), unless
//                             an avatar stands on it
//   Ore             :62-143   one component per ore type over one state machine; a 'mine'
//                             hit adds the hitter to the component's miners and (re)starts
//                             its countdown; with minNumMiners miners the ore is extracted
//                             (rewards, oreWait); the countdown running out forgets the
//                             miners (back to <type>Raw)
//   MineBeam        :147-254  cooldown, hitBeam from the component's update() — i.e. in
//                             BaseSimulation:update, in avatar creation order, AHEAD of
//                             every updater's events in the frame's queue
//   StochasticIntervalEpisodeEnding  component_library.lua:907-948
// The Lua-side variables of an ore live in two hidden planes of the record: the miners
// of its many-miner type as a byte mask, and that type's countdown (a one-miner type is
// extracted by the hit that starts its countdown: it has no state between frames).
#ifndef MP_STEP_COOP_H_
#define MP_STEP_COOP_H_

#include "step_common.h"

namespace stepk {

constexpr int kOreRegs = 10;   // mp_create admits at most 640 ore sites

struct CoopSites { int ore[kOreRegs]; };

__device__ inline CoopSites load_sites(const CoopTables& c, int lane) {
  CoopSites s;
#pragma unroll
  for (int k = 0; k < kOreRegs; ++k) {
    const int i = k * 64 + lane;
    s.ore[k] = i < c.n_ore ? c.ore_cells[i] : -1;
  }
  return s;
}

__device__ inline void step_world(const DevTables& t, const CoopTables& c,
                                  const CoopSites& sites, const World& wd, const Action& act,
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
  int step_type;
  const int alive_state = is_av ? t.alive_state[lane] : 0;
  const int s_wait = c.s_wait, s_raw0 = c.s_raw[0], s_raw1 = c.s_raw[1], s_part1 = c.s_partial[1];
  // FixedRateRegrow of one site at frame `step` of episode `ep`: the state the site's ore
  // takes (0: none).  A12: one draw per piece and updater; A11: registration order, the
  // later updater's setState is processed later and stands.
  auto regrow_of = [&](int i, uint32_t step, uint32_t ep, uint32_t k0, uint32_t k1) -> int {
    int rg = 0;
    if (philox_u53(philox4x32_10((uint32_t)i, RS_REGROW, step, ep, k0, k1)) < c.thr[0]) rg = s_raw0;
    if (philox_u53(philox4x32_10((uint32_t)(c.n_ore + i), RS_REGROW, step, ep, k0, k1)) < c.thr[1])
      rg = s_raw1;
    return rg;
  };

  if (what == 1) {
    // ---- api:start (api_factory.lua:85-102); the episode number is a word of the draw
    // counter (A10; the reference re-seeds with seed + 1, builder.py:177-181)
    const uint32_t k0 = (uint32_t)tail->seed, k1 = (uint32_t)(tail->seed >> 32);
    const uint32_t ep = tail->episode;
    wsync();
    const int gvec = (t.L * HW + 15) >> 4;
    for (int i = lane; i < gvec; i += 64)
      reinterpret_cast<uint4*>(grid)[i] = reinterpret_cast<const uint4*>(t.init_grid)[i];
    wsync();
    for (int i = lane; i < HW; i += 64) { at(c.plane_m, i) = 0; at(c.plane_c, i) = 0; }   // Ore:reset
    if (lane == 0) {
      tail->episode = ep + 1;
      tail->step = 0; tail->frame = 1; tail->done = 0; tail->cont = 1;
      tail->started = 1;
      tail->group_change = 0;
      tail->ctr[2]++;
    }
    if (lane < MP_MAX_PLAYERS) { tail->flag0[lane] = 0; tail->flag1[lane] = 0; }
    wsync();
    apply_map_choices(t, grid, lane, ep, k0, k1);
    spawn_avatars(t, grid, lane, ep, k0, k1, a);
    if (lane < P) push_event(sc, MP_EVENT_AVATAR_STARTED, 0, 0);
    wsync();
    // the grid:update of api:start runs the updaters too: ores may grow at frame 0
    int live = 0;
#pragma unroll
    for (int r = 0; r < kOreRegs; ++r) {
      if (r * 64 >= c.n_ore) break;
      const int cell = sites.ore[r];
      bool grown = false;
      if (cell >= 0 && at(c.ore_layer, cell) == s_wait) {
        const int rg = regrow_of(r * 64 + lane, 0u, ep, k0, k1);
        if (rg != 0 && at(t.avatar_layer, cell) == 0) { at(c.ore_layer, cell) = (uint8_t)rg; grown = true; }
      }
      live += __popcll(__ballot(grown));
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
    const int a_move = act.move, a_turn = act.turn, a_mine = act.fire0, bad = act.bad;

    // ---- BaseSimulation:update, objects in creation order: the avatars ...
    // MineBeam:update (components.lua:228-244): the timer runs down FIRST, and a beam
    // leaves in the very update that brings it to zero
    bool fire = false;
    if (is_av) {
      if (a.ztimer > 0) a.ztimer--;
      if (a_mine == 1 && a.ztimer == 0) { a.ztimer = c.cooldown; fire = true; }
    }
    // ... then the ores.  Ore:update of the many-miner type (components.lua:99-105): its
    // countdown runs out -> the miners are forgotten, setState(<type>Raw) is queued (behind
    // the beams).  And, per site in oreWait, the regrow updaters' draws (the callbacks look
    // at the avatars where they stand NOW: before this frame's moves).
    uint32_t timeout_bits = 0, regrow_lo = 0, regrow_hi = 0;   // regrow: 2 bits a site (0, type 1, type 2)
#pragma unroll
    for (int r = 0; r < kOreRegs; ++r) {
      if (r * 64 >= c.n_ore) break;
      const int cell = sites.ore[r];
      if (cell < 0) continue;
      const int s = at(c.ore_layer, cell);
      const int cd = at(c.plane_c, cell);
      if (cd > 0) {
        at(c.plane_c, cell) = (uint8_t)(cd - 1);
        if (cd == 1) {   // Ore:reset
          at(c.plane_m, cell) = 0;
          if (s != s_wait) timeout_bits |= 1u << r;
        }
      }
      if (s == s_wait) {
        const int rg = regrow_of(r * 64 + lane, (uint32_t)step, ep, k0, k1);
        if (rg != 0 && at(t.avatar_layer, cell) == 0) {
          const uint32_t code = rg == s_raw0 ? 1u : 2u;
          if (r < 16) regrow_lo |= code << (2 * r); else regrow_hi |= code << (2 * (r - 16));
        }
      }
    }
    // ---- updaters (pre-flush state)
    int orders[4];
    step_orders(tail, lane, P, kOrders, (uint32_t)step, ep, k0, k1, orders);
    const int order_move = orders[0];
    int cont = tail->cont;  // StochasticIntervalEpisodeEnding: _t == step + 1
    if (frame >= c.ee_min_frames && (step + 1) % c.ee_interval == 0)
      if (philox_u53(philox4x32_10(0u, RS_EPISODE_END, (uint32_t)step, ep, k0, k1)) < c.thr[2]) cont = 0;
    // beam sprites of the previous frame disappear (grid:update start)
    clear_bytes(grid, c.beam_layer * HW, HW, lane);
    wsync();

    // ---- flush 1.  The beams come first (queued by the avatars' update()): every beam
    // sees the ores as the frame found them — a setState a hit queues is processed in
    // flush 2 — so the footprints are evaluated together; what a hit DOES (miners,
    // countdown, rewards, events) follows the queue: beam by beam in avatar order, footprint
    // order inside a beam, against the components' variables as the earlier hits left them.
    fire_beams(t, wd, tail, a, fire, beam_lane(c.shape, lane), c.hit, false,
               c.beam_layer, c.s_beam, false,
               // Ore:onHit (components.lua:113-143): a raw or partial ore reacts and stops the beam
               [&](int s, int) { return (s == s_raw0 || s == s_raw1 || s == s_part1) ? 3 : 0; },
               [&](int b0, int per, int nc, bool reached, int cell, bool touched) {
                 (void)per; (void)reached;
                 for (unsigned long long m = __ballot(touched); m != 0ull; m &= m - 1ull) {
                   const int src = __ffsll((long long)m) - 1;      // lanes are (beam, cell) in queue order
                   const int owner = b0 + (int)(((uint32_t)src * c.shape.magic) >> 16);   // src / nc
                   const int hc = rdlane(cell, src);
                   const int s = at(c.ore_layer, hc);
                   if (s == s_raw0) {
                     // the one-miner type: mined and extracted by the same hit
                     if (lane == owner) {
                       a.reward += c.reward[owner * 4 + 0];
                       a.reward += c.reward[owner * 4 + 2];
                       push_event(sc, MP_EVENT_MINING, owner + 1, 1);
                       push_event(sc, MP_EVENT_EXTRACTION, owner + 1, 1);
                       wd.mark[hc] = (uint8_t)s_wait;
                     }
                   } else {
                     // Ore:addMiner, MineBeam:processRoleMineEvent
                     const uint32_t mask = (uint32_t)at(c.plane_m, hc) | (1u << owner);
                     if (lane == owner) {
                       a.reward += c.reward[owner * 4 + 1];
                       push_event(sc, MP_EVENT_MINING, owner + 1, 2);
                     }
                     if (__popc(mask) == c.min_miners1) {
                       // extraction: every miner is paid, every ordered pair reported; Ore:reset
                       if (is_av && ((mask >> lane) & 1u)) {
                         a.reward += c.reward[lane * 4 + 3];
                         push_event(sc, MP_EVENT_EXTRACTION, lane + 1, 2);
                         for (int o2 = 0; o2 < P; ++o2)
                           if (o2 != lane && ((mask >> o2) & 1u))
                             push_event(sc, MP_EVENT_EXTRACTION_PAIR, lane + 1, ((o2 + 1) << 2) | 2);
                       }
                       if (lane == 0) { at(c.plane_m, hc) = 0; at(c.plane_c, hc) = 0; wd.mark[hc] = (uint8_t)s_wait; }
                     } else if (lane == 0) {
                       at(c.plane_m, hc) = (uint8_t)mask;
                       at(c.plane_c, hc) = (uint8_t)c.window1;
                       wd.mark[hc] = (uint8_t)s_part1;
                     }
                   }
                   wsync();   // the next hit reads what this one wrote
                 }
               });
    // ... then what the ores' update() and the regrow updaters queued
#pragma unroll
    for (int r = 0; r < kOreRegs; ++r) {
      if (r * 64 >= c.n_ore) break;
      const int cell = sites.ore[r];
      if (cell < 0) continue;
      if ((timeout_bits >> r) & 1u) at(c.ore_layer, cell) = (uint8_t)s_raw1;
      const uint32_t code = r < 16 ? (regrow_lo >> (2 * r)) & 3u : (regrow_hi >> (2 * (r - 16))) & 3u;
      if (code) at(c.ore_layer, cell) = (uint8_t)(code == 1u ? s_raw0 : s_raw1);
    }
    // ... then the moves, in visiting order (nothing reacts to a contact in this level)
    (void)resolve_moves(t, wd, a, a_move, a_turn, order_move, alive_state);

    // ---- flush 2: the setStates the hits queued (the last one per ore stands)
    int live = 0;
#pragma unroll
    for (int r = 0; r < kOreRegs; ++r) {
      if (r * 64 >= c.n_ore) break;
      const int cell = sites.ore[r];
      if (cell >= 0) {
        const int mk = wd.mark[cell];
        if (mk != 0) { at(c.ore_layer, cell) = (uint8_t)mk; wd.mark[cell] = 0; }
      }
      live += __popcll(__ballot(cell >= 0 && at(c.ore_layer, cell >= 0 ? cell : 0) != s_wait));
    }
    wsync();
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
  // READY_TO_SHOOT reads the MineBeam (ReadyToShootObservation.zapperComponent)
  finish(t, wd, tail, a, 0.0, c.cooldown, step_type, out, kOrders);
}

}  // namespace stepk

#endif  // MP_STEP_COOP_H_
