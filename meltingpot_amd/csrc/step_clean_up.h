// step_clean_up.h — one environment step (or episode start) of one clean_up
// world, by one wavefront, on the world's record in LDS.
//
// Replaces, for the clean_up substrate, the reference's per-step path
//   api:advance            lua/modules/api_factory.lua:104-111
//   BaseSimulation:update  lua/modules/base_simulation.lua:476-486
//   grid:update(random)    dmlab2d (documented cycle: docs/advanced.md:33-52)
// and the Lua component callbacks it drives (cited at each block below), and
// the episode start path api:start (api_factory.lua:85-102,
// base_simulation.lua:396-471).
//
// Execution shape.  The wave's world record (grid planes + tail, ≈6 KB) sits in
// LDS; it is stepped there and streamed back.  Nothing in the step is resolved
// by a serial lane:
//   * lane p (< P) owns avatar p — position, orientation, timers, reward live in
//     its registers for the whole step;
//   * site work (122 AppleGrow draws, 147 dirt sites, 167 water pieces, plane
//     clears) is spread over the 64 lanes, with ballots for set selection; the
//     site lists (cell per site) are held in registers (CleanUpSites), loaded
//     once per wave next to the first record;
//   * the frame's shuffled visiting orders (A1) are drawn one Philox call per
//     lane and applied on the scalar unit;
//   * moves and respawns are resolved in visiting order with one ballot per
//     avatar ("is any live avatar standing on my target?") instead of grid
//     reads, so the ordered phase touches no memory at all;
//   * beams: lane (b, j) evaluates footprint cell j of avatar b's beam; a
//     ballot of the "stops the beam" predicate against a per-cell predecessor
//     mask gives every cell's reached/not-reached in one step (63 lanes for 7
//     avatars x 9 cells).  Beams never change state inside the flush (their
//     effects are queued to the next flush), so all beams evaluate at once.
// The substrate-independent pieces (avatars, moves, beams, respawns) live in
// step_common.h.  v1 ran the ordered phase on lane 0 against LDS and took
// ~67 us for 4096 worlds (profiles/r01_v1_baseline.md): pure LDS latency.
#ifndef MP_STEP_CLEAN_UP_H_
#define MP_STEP_CLEAN_UP_H_

#include "step_common.h"

#ifdef MP_STEP_TIMING   // developer build: per-phase cycle stamps of one world
#include <stdio.h>
#define TSTAMP(i) ts_[i] = __builtin_readcyclecounter()
#else
#define TSTAMP(i)
#endif

namespace stepk {

constexpr int kSiteRegs = 4;   // mp_create admits at most 64 * kSiteRegs sites per list

// The cells of this lane's sites (site k * 64 + lane), -1 beyond the list.
struct CleanUpSites { int apple[kSiteRegs], dirt[kSiteRegs], water[kSiteRegs]; };

__device__ inline CleanUpSites load_sites(const CleanUpTables& c, int lane) {
  CleanUpSites s;
#pragma unroll
  for (int k = 0; k < kSiteRegs; ++k) {
    const int i = k * 64 + lane;
    s.apple[k] = i < c.n_apple ? c.apple_cells[i] : -1;
    s.dirt[k] = i < c.n_dirt ? c.dirt_cells[i] : -1;
    s.water[k] = i < c.n_water ? c.water_cells[i] : -1;
  }
  return s;
}

__device__ inline void step_world(const DevTables& t, const CleanUpTables& c,
                                  const CleanUpSites& sites, const World& wd, const Action& act,
                                  const StepArgs& args) {
  const int lane = wd.lane, w = wd.w;
  const StepOutputs& out = args.out;
#ifdef MP_STEP_TIMING
  unsigned long long ts_[12] = {0};
#endif
  TSTAMP(0);
  Scratch* sc = wd.sc;
  uint8_t* mark = wd.mark;      // dirt cells hit by a clean beam
  uint8_t* grid = wd.rec;
  WorldTail* tail = reinterpret_cast<WorldTail*>(wd.rec + t.grid_pad);
  const int P = t.P, HW = t.H * t.W, W = t.W;
  const bool is_av = lane < P;
  auto at = [&](int layer, int cell) -> uint8_t& { return grid[layer * HW + cell]; };

  const OrderStreams kOrders = {RS_SHUFFLE_MOVE, RS_SHUFFLE_ZAP, RS_SHUFFLE_RESPAWN, 0, 3};   // the updater groups shuffled per frame (A1)

  const int what = dispatch(t, tail, lane, w, args.reset_mask, args.mode, args.auto_reset, out);
  if (what == 0) return;
  TSTAMP(1);

  Av a;
  double aux0 = 0.0;
  double dbg_cleaned = 0.0, dbg_ate = 0.0, dbg_zapped = 0.0, dbg_others_ate = 0.0;
  int step_type;
  const int alive_state = is_av ? t.alive_state[lane] : 0;
  double* zmat = out.zap_matrix ? out.zap_matrix + (size_t)w * P * P : nullptr;
  if (zmat) for (int i = lane; i < P * P; i += 64) zmat[i] = 0.0;

  if (what == 1) {
    // ---- api:start(episode, seed) (api_factory.lua:85-102).  The reference
    // rebuilds with seed + 1 on every reset (builder.py:177-181); here the
    // episode number is a word of the draw counter (A10).
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
      tail->aux_count = c.n_dirt_init;  // DirtTracker:postStart (:103-116)
      tail->group_change = 0;
      tail->ctr[2]++;
    }
    if (lane < MP_MAX_PLAYERS) { tail->flag0[lane] = 0; tail->flag1[lane] = 0; }
    wsync();
    apply_map_choices(t, grid, lane, ep, k0, k1);
    spawn_avatars(t, grid, lane, ep, k0, k1, a);
    if (is_av) push_event(sc, MP_EVENT_AVATAR_STARTED, 0, 0);
    // Animation:postStart with randomStartFrame (component_library.lua:1064):
    // the queued setState is flushed by the grid:update at api_factory.lua:101.
#pragma unroll
    for (int q = 0; q < kSiteRegs; ++q) {
      if (sites.water[q] < 0) continue;
      const uint32_t k = philox_bounded(
          philox4x32_10((uint32_t)(q * 64 + lane), RS_ANIM_START, 0u, ep, k0, k1), 4u);
      at(c.water_layer, sites.water[q]) = (uint8_t)(c.s_water_packed >> (8u * k));
    }
    step_type = 0;
  } else {
    // ================= api:advance =================
    const uint32_t k0 = (uint32_t)tail->seed, k1 = (uint32_t)(tail->seed >> 32);
    const uint32_t ep = tail->episode - 1;
    const int step = tail->step + 1, frame = tail->frame;
    const int dirt_count0 = tail->aux_count;
    // (growth threshold of this frame: the only table read of the step that
    // depends on the record; issued first, used after the DirtSpawner)
    const uint64_t apple_thr = c.apple_thr[dirt_count0];
    int flag0 = 0, flag1 = 0;  // GlobalData cleaned / ate ThisStep (previous step's flush)
    load_avatars(tail, lane, a);
    if (lane < MP_MAX_PLAYERS) { flag0 = tail->flag0[lane]; flag1 = tail->flag1[lane]; }
    wsync();
    auto draw = [&](int stream, uint32_t index) {
      return philox4x32_10(index, (uint32_t)stream, (uint32_t)step, ep, k0, k1);
    };
    // api:discreteActions (api_factory.lua:81) + the ACTION_SET lookup of
    // discrete_action_wrapper.py:97-109; Avatar:preUpdate resets the reward.
    const int a_move = act.move, a_turn = act.turn, a_zap = act.fire0, a_clean = act.fire1,
              bad = act.bad;
    // beam sprites of the previous frame disappear (grid:update start)
    clear_bytes(grid, c.zap.layer * HW, HW, lane);
    clear_bytes(grid, c.clean_layer * HW, HW, lane);
    TSTAMP(2);

    // ---- BaseSimulation:update: DirtSpawner:update (clean_up/components.lua:329-340)
    if (step > c.dirt_delay) {
      const Philox4 d = draw(RS_DIRT_SPAWN, 0);
      if (philox_u53(d) < c.thr_dirt_spawn) {
        unsigned long long m[kSiteRegs];
        int n = 0;
#pragma unroll
        for (int q = 0; q < kSiteRegs; ++q) {
          m[q] = __ballot(sites.dirt[q] >= 0 &&
                          at(c.dirt_wait_layer, sites.dirt[q] >= 0 ? sites.dirt[q] : 0) == c.s_dirt_wait);
          n += __popcll(m[q]);
        }
        if (n > 0) {  // random:choice(set.toSortedList(potential))
          const int k = (int)philox_bounded(d, (uint32_t)n);
          const int site = kth_site(m[0], m[1], m[2], m[3], k);
          // first event of the flush: the DirtSpawner setState
          const int q = site >> 6;
          const int cell = rdlane(q == 0 ? sites.dirt[0] : q == 1 ? sites.dirt[1]
                                  : q == 2 ? sites.dirt[2] : sites.dirt[3], site & 63);
          if (lane == 0 && at(c.dirt_layer, cell) == 0) {
            at(c.dirt_wait_layer, cell) = 0;
            at(c.dirt_layer, cell) = (uint8_t)c.s_dirt;
          }
        }
      }
    }
    TSTAMP(3);
    // ---- AppleGrow:update (clean_up/components.lua:64-80): one draw per
    // potential apple; the probability depends on the dirt count only (as it
    // was when update() ran, i.e. before this frame's events).
    // (threshold 0 = the river is too dirty for anything to grow — every step of
    // random play on the stock map: no draw can be below it, none is made)
    if (apple_thr != 0) {
#pragma unroll
      for (int q = 0; q < kSiteRegs; ++q) {
        const int cell = sites.apple[q];
        if (cell < 0) continue;
        if (philox_u53(draw(RS_APPLE_GROW, (uint32_t)(q * 64 + lane))) < apple_thr)
          if (at(c.apple_layer, cell) == 0) at(c.apple_layer, cell) = (uint8_t)c.s_apple;
      }
    }

    TSTAMP(4);
    // ---- updaters, priority descending (updater_registry.lua:166-173); they
    // read the pre-flush state and queue events.
    int orders[4];
    step_orders(tail, lane, P, kOrders, (uint32_t)step, ep, k0, k1, orders);
    const int order_move = orders[0], order_zap = orders[1], order_resp = orders[2];
    // (the Cleaner order, RS_SHUFFLE_CLEAN, has no observable effect: beams do
    // not change state inside the flush and cleanHit carries no reward)
    bool fire_zap = false, fire_clean = false, want_respawn = false;
    if (is_av) {
      // 140 Zapper zap (avatar_library.lua:613-636)
      if (a.alive && c.zap.cooldown >= 0) {
        if (a.ztimer > 0) a.ztimer--;
        else if (a_zap == 1) { a.ztimer = c.zap.cooldown; fire_zap = true; }
      }
      // 140 Cleaner clean (clean_up/components.lua:201-224)
      if (a.alive && c.clean_cooldown >= 0) {
        if (a.ctimer > 0) a.ctimer--;
        else if (a_clean == 1) { a.ctimer = c.clean_cooldown; fire_clean = true; }
      }
      // 135 Zapper respawn: state = waitState, startFrame = framesTillRespawn
      // (avatar_library.lua:638-649)
      want_respawn = !a.alive && (frame - a.achange) >= c.zap.respawn_frames;
    }
    // 100 Animation (component_library.lua:1070-1094)
    const bool water_advance = (frame - tail->group_change) >= c.anim_frames;
    // 100 StochasticIntervalEpisodeEnding (component_library.lua:927-948):
    // _t was incremented by update() this step, so _t == step + 1.
    int cont = tail->cont;
    if (frame >= c.ee_min_frames && (step + 1) % c.ee_interval == 0)
      if (philox_u53(draw(RS_EPISODE_END, 0)) < c.thr_episode_end) cont = 0;
    // 4 AllNonselfCumulants.getCumulants (:535-545), 2 GlobalData.resetCumulants
    {
      const int total = __popcll(__ballot(is_av && flag0 != 0));
      aux0 = (double)(total - (flag0 != 0 ? 1 : 0));
      const int total_ate = __popcll(__ballot(is_av && flag1 != 0));
      dbg_others_ate = (double)(total_ate - (flag1 != 0 ? 1 : 0));
    }
    int cleaned = 0, ate = 0;  // this frame's GlobalData flags

    TSTAMP(5);
    // ---- flush 1: queued events in FIFO order (docs/advanced.md:43-52)
    const bool wants = resolve_moves(t, wd, a, a_move, a_turn, order_move, alive_state);
    // onContact 'avatar' enter on the destination — or, for a blocked move, on
    // the cell the avatar stays in (A3b): Edible:onEnter + Taste:consumed
    // (clean_up/components.lua:390-408,446-455); apple -> appleWait next flush.
    int ate_cell = -1;
    if (wants && at(c.apple_layer, a.y * W + a.x) == c.s_apple) {
      a.reward += c.eat_reward; ate = 1; ate_cell = a.y * W + a.x;
      push_event(sc, MP_EVENT_EDIBLE_CONSUMED, lane + 1, 0);
    }
    wsync();

    TSTAMP(6);
    int nzapped = 0, ncleaned = 0;
    fire_beams(t, wd, tail, a, fire_zap, beam_lane(c.zap.shape, lane), c.zap.hit, true,
               c.zap.layer, c.zap.s_hit, c.zap.remove_hit != 0,
               [](int, int) { return 0; },
               [](int, int, int, bool, int, bool) {}, -1, zmat, &nzapped);
    zap_rewards(t, sc, lane, a, fire_zap, order_zap, c.zap.shape.n, c.zap.penalty,
                c.zap.reward);
    TSTAMP(7);
    const BeamLane clean_lane = beam_lane(c.clean_shape, lane);
    fire_beams(t, wd, tail, a, fire_clean, clean_lane, c.clean_hit, false,
               c.clean_layer, c.s_clean_hit, false,
               // DirtCleaning:onHit (clean_up/components.lua:141-157)
               [&](int s, int) { return s == c.s_dirt ? 3 : 0; },
               [&](int b0, int per, int nc, bool reached, int cell, bool dhit) {
                 (void)reached;
                 if (dhit) {
                   mark[cell] = 1;  // dirt -> dirtWait in the next flush
                   push_event(sc, MP_EVENT_PLAYER_CLEANED, b0 + clean_lane.bl + 1, 0);
                 }
                 const unsigned long long db = __ballot(dhit);
                 if (db == 0) return;
                 if (lane == 0) tail->ctr[5] += __popcll(db);
                 // Cleaner:setCumulant for the beam's owner: player_cleaned + 1 per
                 // dirt hit, GlobalData:setCleanedThisStep
                 if (lane >= b0 && lane < b0 + per && lane < P) {
                   const int n = __popcll((db >> ((lane - b0) * nc)) & ((1ull << nc) - 1ull));
                   ncleaned += n;
                   if (n) cleaned = 1;
                 }
               });

    TSTAMP(8);
    const int rcell = resolve_respawns(t, wd, tail, a, want_respawn, order_resp, alive_state,
                                       (uint32_t)step, frame, ep, k0, k1);
    if (rcell >= 0 && at(c.apple_layer, rcell) == c.s_apple) {  // placed on a live apple
      a.reward += c.eat_reward; ate = 1; ate_cell = rcell;
      push_event(sc, MP_EVENT_EDIBLE_CONSUMED, lane + 1, 0);
    }
    // water Animation setStates: the last events of flush 1
    if (water_advance) {
#pragma unroll
      for (int q = 0; q < kSiteRegs; ++q) {
        const int cell = sites.water[q];
        if (cell < 0) continue;
        // (the four frame states are packed into one word: a per-lane index into
        // the kernel argument would be a memory access)
        const uint32_t s = at(c.water_layer, cell);
        uint32_t k = 0;
#pragma unroll
        for (uint32_t f = 1; f < 4; ++f) if (s == ((c.s_water_packed >> (8u * f)) & 255u)) k = f;
        at(c.water_layer, cell) = (uint8_t)(c.s_water_packed >> (8u * ((k + 1u) & 3u)));
      }
    }
    wsync();

    TSTAMP(9);
    // ---- flush 2: setStates queued by the callbacks of flush 1
    if (ate_cell >= 0) at(c.apple_layer, ate_cell) = 0;   // apple -> appleWait (off-grid)
    apply_zapped(t, wd, a, rcell >= 0, frame);
    wsync();
    int dirt_count = 0;
    {
      // sweep: apply the marked dirt -> dirtWait transitions (several beams may
      // have hit one cell) and recount
      // (RiverMonitor / DirtTracker:onStateChange, clean_up/components.lua:118-129)
#pragma unroll
      for (int q = 0; q < kSiteRegs; ++q) {
        if (q * 64 >= c.n_dirt) break;
        bool dirty = false;
        const int cell = sites.dirt[q];
        if (cell >= 0) {
          if (mark[cell]) {                                // dirt -> dirtWait
            mark[cell] = 0;
            if (at(c.dirt_layer, cell) == c.s_dirt && at(c.dirt_wait_layer, cell) == 0) {
              at(c.dirt_layer, cell) = 0;
              at(c.dirt_wait_layer, cell) = (uint8_t)c.s_dirt_wait;
            }
          }
          dirty = at(c.dirt_layer, cell) == c.s_dirt;
        }
        dirt_count += __popcll(__ballot(dirty));
      }
    }
    const unsigned long long badb = __ballot(bad != 0);
    const int done = !(cont && step < t.max_frames);  // api_factory.lua:107-110
    if (lane == 0) {
      tail->step = step;
      tail->frame = frame + 1;
      tail->cont = cont;
      tail->done = done;
      tail->aux_count = dirt_count;
      if (water_advance) tail->group_change = frame;
      tail->ctr[0]++; tail->ctr[1] += (uint32_t)P; tail->ctr[7] += __popcll(badb);
    }
    if (lane < MP_MAX_PLAYERS) { tail->flag0[lane] = (uint8_t)cleaned; tail->flag1[lane] = (uint8_t)ate; }
    step_type = done ? 2 : 1;
    dbg_cleaned = (double)ncleaned; dbg_ate = (double)ate; dbg_zapped = (double)nzapped;
  }

  TSTAMP(10);
  // debug metrics (clean_up.py:751-784): PLAYER_CLEANED, PLAYER_ATE_APPLE,
  // NUM_OTHERS_PLAYER_ZAPPED_THIS_STEP, NUM_OTHERS_WHO_ATE_THIS_STEP
  if (is_av) {
    const size_t o = (size_t)w * P + lane;
    if (out.dbg[0]) out.dbg[0][o] = dbg_cleaned;
    if (out.dbg[1]) out.dbg[1][o] = dbg_ate;
    if (out.dbg[2]) out.dbg[2][o] = dbg_zapped;
    if (out.dbg[3]) out.dbg[3][o] = dbg_others_ate;
  }
  // NUM_OTHERS_WHO_CLEANED_THIS_STEP is the substrate metric
  // (component_library.lua:786-803)
  finish(t, wd, tail, a, aux0, c.zap.cooldown, step_type, out, kOrders);
  TSTAMP(11);
#ifdef MP_STEP_TIMING
  if (lane == 0 && (w == 7 || w == 2000) && what == 2)
    printf("w %d: dispatch %llu clear %llu dirt %llu apple %llu upd %llu moves %llu zap %llu clean %llu resp %llu flush2 %llu finish %llu total %llu\n",
           w, ts_[1] - ts_[0], ts_[2] - ts_[1], ts_[3] - ts_[2], ts_[4] - ts_[3], ts_[5] - ts_[4],
           ts_[6] - ts_[5], ts_[7] - ts_[6], ts_[8] - ts_[7], ts_[9] - ts_[8], ts_[10] - ts_[9],
           ts_[11] - ts_[10], ts_[11] - ts_[0]);
#endif
}

}  // namespace stepk

#endif  // MP_STEP_CLEAN_UP_H_
