// step_territory.h — one environment step (or episode start) of one territory
// world by one wavefront (shape: step_clean_up.h / step_common.h).
//
// Substrate rules restated here (reference: configs/substrates/territory.py,
// territory__rooms.py; lua/levels/territory/components.lua;
// lua/modules/avatar_library.lua):
//   Resource           components.lua:51-210   claim / zap damage / self repair /
//                                              reward while claimed / release
//   ResourceClaimer    :214-276                claimBeam_<i>, passes resources
//   Paintbrush         :362-412                directionHit<i>, length 1, every frame
//   RewardIndicator    :279-313                dry-paint overlay while rewarding
//   GraduatedSanctionsMarking  avatar_library.lua:948-1121  level 1 -> freeze,
//                                              level 2 -> removal, recovery
//   Avatar / Zapper timed freeze, zap prevention, scheduled removal
//                                              avatar_library.lua:334-355,704-726
//
// Per-resource variables live in three hidden grid planes behind the L render
// planes (one byte per map cell, so a resource lane addresses them by cell):
//   plane A  health (bits 0-1) | rewardingStatus active (bit 2) | claimedBy+1 (bits 3-7)
//   plane B  framesSinceZapped + 1, saturating (0 = nil)
//   plane C  frames since the resource's last state change, saturating
//            (grid:frames(piece) for the startFrame tests: 25 and 5)
// The marking overlay piece of an avatar (connected to it, A14) is a byte on
// the superOverlay plane at the avatar's cell; its state (wait / level_1 /
// level_2) is tail->flag0[p].  It outlives its avatar by one flush — and for good
// if a zap hits it in exactly that flush (the queued _setLevel lands after the
// queued wait state): such an orphan stays where the avatar died, still reacts
// to zaps and blocks other avatars' markings, as the restated Lua does.
//
// Unlike clean_up, zap beams change state inside the flush (Resource._health is
// updated in the onHit callback and decides whether the NEXT beam is stopped),
// so zap beams are evaluated one at a time in visiting order; each beam is still
// lane-parallel over its footprint.  Brush and claim beams only ever request
// state changes for the next flush; they evaluate all at once and the
// last-caller-wins rules of Resource:_claim are resolved with LDS atomicMax on
// (event sequence number, player).
#ifndef MP_STEP_TERRITORY_H_
#define MP_STEP_TERRITORY_H_

#include "step_common.h"

#ifdef MP_STEP_TIMING   // developer build: per-phase cycle stamps of one world
#include <stdio.h>
#ifndef TSTAMP
#define TSTAMP(i) ts_[i] = __builtin_readcyclecounter()
#endif
#else
#ifndef TSTAMP
#define TSTAMP(i)
#endif
#endif

namespace stepk {

constexpr int kResPerLane = 4;   // mp_create admits at most 64 * kResPerLane resources

// this lane's resources (i = k * 64 + lane): every rule loop walks them, and the
// table lives in global memory
struct TerritorySites { int res[kResPerLane]; };

__device__ inline TerritorySites load_sites(const TerritoryTables& c, int lane) {
  TerritorySites s;
#pragma unroll
  for (int k = 0; k < kResPerLane; ++k)
    s.res[k] = k * 64 + lane < c.n_res ? c.res_cells[k * 64 + lane] : -1;
  return s;
}

// LDS behind the marks of a scratch slot (16-byte aligned).
struct TrScratch {
  // small tables of TerritoryTables that are indexed per lane (a dynamically
  // indexed kernel argument is a ~500-cycle constant-memory load each time);
  // written once per scratch slot by init_extra
  int8_t owner[256];                  // state -> player whose claimed_by state it is, or -1
  uint8_t s_claimed[MP_MAX_PLAYERS], s_dry[MP_MAX_PLAYERS], s_claim_hit[MP_MAX_PLAYERS];
  uint8_t hit_claim[MP_MAX_PLAYERS], s_brush[MP_MAX_PLAYERS][4];
  // per step
  int32_t reward_count[MP_MAX_PLAYERS];
  uint8_t av_ori[MP_MAX_PLAYERS];
  int16_t mark_cell[MP_MAX_PLAYERS];  // cell of avatar p's marking overlay, or -1
  // followed by uint16_t lastcall[H*W], lastdiff[H*W]: per cell (of a resource),
  // tag of the last _claim call in this flush / of the last one by a non-owner
  // (0 = none; indexed by cell so that a beam lane needs no cell -> resource table)
};
static_assert(sizeof(TrScratch) % 16 == 0, "TrScratch keeps lastcall[] aligned");

__host__ __device__ inline int extra_bytes(const TerritoryTables& c) {
  return ((int)sizeof(TrScratch) + ((c.map_cells + 1) & ~1) * 4 + 15) & ~15;
}

// Once per scratch slot: the read-only part of TrScratch.
__device__ inline void init_extra(const DevTables& t, const TerritoryTables& c, uint8_t* extra,
                                  int lane) {
  TrScratch* ts = reinterpret_cast<TrScratch*>(extra);
  for (int i = lane; i < 256; i += 64) ts->owner[i] = -1;
  {
    uint32_t* calls = reinterpret_cast<uint32_t*>(ts + 1);   // lastcall + lastdiff
    for (int i = lane; i < ((c.map_cells + 1) & ~1); i += 64) calls[i] = 0u;
  }
  wsync();
  for (int p = 0; p < t.P_pack; ++p) {   // (uniform index: scalar reads of the arguments)
    if (lane == 0) {
      ts->owner[c.s_claimed[p]] = (int8_t)p;
      ts->s_claimed[p] = (uint8_t)c.s_claimed[p];
      ts->s_dry[p] = (uint8_t)c.s_dry[p];
      ts->s_claim_hit[p] = (uint8_t)c.s_claim_hit[p];
      ts->hit_claim[p] = (uint8_t)c.hit_claim[p];
    }
    for (int d = 0; d < 4; ++d)
      if (lane == 0) ts->s_brush[p][d] = (uint8_t)c.s_brush[p][d];
  }
  wsync();
}

__device__ inline void step_world(const DevTables& t, const TerritoryTables& c,
                                  const TerritorySites& sites, const World& wd,
                                  const Action& act, const StepArgs& args) {
  const int lane = wd.lane, w = wd.w;
  const StepOutputs& out = args.out;
#ifdef MP_STEP_TIMING
  unsigned long long ts_[10] = {0};
#endif
  TSTAMP(0);
  Scratch* sc = wd.sc;
  const int P = t.P, HW = t.H * t.W, W = t.W;
  uint8_t* mark = wd.mark;  // bit0 release, bit1 destroyed this frame
  TrScratch* ts = reinterpret_cast<TrScratch*>(wd.extra);
  uint16_t* lastcall = reinterpret_cast<uint16_t*>(ts + 1);
  uint16_t* lastdiff = lastcall + ((c.map_cells + 1) & ~1);
  uint8_t* grid = wd.rec;
  WorldTail* tail = reinterpret_cast<WorldTail*>(wd.rec + t.grid_pad);
  const bool is_av = lane < P;
  auto at = [&](int layer, int cell) -> uint8_t& { return grid[layer * HW + cell]; };

  const OrderStreams kOrders = {RS_SHUFFLE_MOVE, RS_SHUFFLE_ZAP, RS_SHUFFLE_BRUSH, RS_SHUFFLE_CLAIM, 4};   // the updater groups shuffled per frame (A1)

  const int what = dispatch(t, tail, lane, w, args.reset_mask, args.mode, args.auto_reset, out);
  if (what == 0) return;
  TSTAMP(1);
  const bool is_reset = what == 1;
  const int (&rcell)[kResPerLane] = sites.res;
  const int alive_state = is_av ? t.alive_state[lane] : 0;

  Av a;
  int freeze = 0, removal = 0, mov_allowed = 1, disallow = 0, nozap = 0, level = 1, tsince = 0;
  int mstate = 0;  // marking piece: 0 wait (off-grid), 1 level_1, 2 level_2
  int a_move = 0, a_turn = 0, a_zap = 0, a_claim = 0, bad = 0;
  bool remove_now = false;
  uint32_t k0, k1, ep;
  int step, frame;

#pragma unroll
  for (int k = 0; k < kResPerLane; ++k)
    if (rcell[k] >= 0) { lastcall[rcell[k]] = 0; lastdiff[rcell[k]] = 0; }
  if (lane < MP_MAX_PLAYERS) ts->reward_count[lane] = 0;

  if (is_reset) {
    // ---- api:start (api_factory.lua:85-102); seed + #earlier resets (builder.py:177-181)
    k0 = (uint32_t)tail->seed; k1 = (uint32_t)(tail->seed >> 32);
    ep = tail->episode;
    step = 0; frame = 0;
    wsync();
    const int gvec = (t.L * HW + 15) >> 4;
    for (int i = lane; i < gvec; i += 64)
      reinterpret_cast<uint4*>(grid)[i] = reinterpret_cast<const uint4*>(t.init_grid)[i];
    wsync();
    apply_map_choices(t, grid, lane, ep, k0, k1);
    for (int i = lane; i < HW; i += 64) {  // hidden planes (and the pad bytes copied above)
      at(c.plane_a, i) = 0; at(c.plane_b, i) = 0; at(c.plane_c, i) = 0;
    }
    wsync();
    // Resource:reset; a resource that is not in this episode's map keeps health 0
    // ("absent": every rule below skips it)
    for (int i = lane; i < c.n_res; i += 64)
      if (at(c.res_layer, c.res_cells[i]) != 0)
        at(c.plane_a, c.res_cells[i]) = (uint8_t)c.initial_health;
    if (lane == 0) {
      tail->episode = ep + 1;
      tail->done = 0; tail->cont = 1; tail->started = 1;
      tail->group_change = 0;
      tail->ctr[2]++;
    }
    if (lane < MP_MAX_PLAYERS) { tail->flag0[lane] = 0; tail->flag1[lane] = 0; }
    wsync();
    spawn_avatars(t, grid, lane, ep, k0, k1, a);
    // GraduatedSanctionsMarking:postStart (avatar_library.lua:1034-1049): the
    // marking is set to level_1, teleported onto its avatar and connected.
    if (is_av) {
      mstate = 1; at(c.mark_layer, a.y * W + a.x) = (uint8_t)c.s_mark[0];
      push_event(sc, MP_EVENT_AVATAR_STARTED, 0, 0);
      push_event(sc, MP_EVENT_SET_SANCTIONING_LEVEL, lane + 1, 1);  // _setLevel in postStart
    }
    // (no BaseSimulation:update at start: only the grid:update below runs)
  } else {
    // ================= api:advance =================
    k0 = (uint32_t)tail->seed; k1 = (uint32_t)(tail->seed >> 32);
    ep = tail->episode - 1;
    step = tail->step + 1; frame = tail->frame;
    load_avatars(tail, lane, a);
    if (lane < MP_MAX_PLAYERS) {
      freeze = tail->freeze[lane]; removal = tail->removal[lane];
      mov_allowed = tail->aflags[lane] & 1; disallow = (tail->aflags[lane] >> 1) & 1;
      nozap = tail->nozap[lane]; level = tail->level[lane]; tsince = tail->tsince[lane];
      mstate = tail->flag0[lane];
    }
    a_move = act.move; a_turn = act.turn; a_zap = act.fire0; a_claim = act.fire1; bad = act.bad;
    wsync();
    // ---- BaseSimulation:update, objects in creation order
    if (is_av) {
      // Avatar:update (avatar_library.lua:334-355)
      if (freeze == 1) mov_allowed = 1;
      if (freeze > 0) freeze--;
      remove_now = removal == 1;
      if (removal > 0) removal--;
      // Zapper:update (avatar_library.lua:713-726)
      if (disallow) a.ztimer = c.zap.cooldown + 1;
      const int old = nozap;
      if (nozap > 0) nozap--;
      if (old == 1) disallow = 0;
    }
    #pragma unroll
    for (int k = 0; k < kResPerLane; ++k) {
      const int cell = rcell[k], i = k * 64 + lane;
      if (cell < 0) continue;
      int A = at(c.plane_a, cell), B = at(c.plane_b, cell);
      int health = A & 3;
      // Resource:update (territory/components.lua:193-206)
      if (health < c.initial_health && health > 0) {
        int dmg = c.s_dmg_damaged;
        if (B > 0 && B - 1 >= c.repair_delay &&
            philox_u53(philox4x32_10((uint32_t)i, RS_SELF_REPAIR, (uint32_t)step, ep, k0, k1)) <
                c.thr_repair) {
          health++;
          if (health == c.initial_health) dmg = c.s_dmg_inactive;
        }
        if (B < 255) B++;
        at(c.dmg_layer, cell) = (uint8_t)dmg;
        at(c.plane_b, cell) = (uint8_t)B;
        at(c.plane_a, cell) = (uint8_t)((A & ~3) | health);
      }
      // RewardIndicator:update (:299-308)
      const int owner = ts->owner[at(c.res_layer, cell)];
      at(c.ind_layer, cell) = (uint8_t)((((A >> 2) & 1) && owner >= 0) ? ts->s_dry[owner] : 0);
    }
  }
  auto draw = [&](int stream, uint32_t index) {
    return philox4x32_10(index, (uint32_t)stream, (uint32_t)step, ep, k0, k1);
  };
  // beam sprites of the previous frame disappear (grid:update start)
  clear_bytes(grid, c.zap.layer * HW, HW, lane);
  clear_bytes(grid, c.brush_layer * HW, HW, lane);
  clear_bytes(grid, c.claim_layer * HW, HW, lane);
  wsync();

  TSTAMP(2);
  // ---- updaters, priority descending; they read the pre-flush state
  int orders[4];
  step_orders(tail, lane, P, kOrders, (uint32_t)step, ep, k0, k1, orders);
  const int order_move = orders[0], order_zap = orders[1], order_brush = orders[2],
            order_claim = orders[3];
  int rank_brush = 0, rank_claim = 0;  // inverse permutations
  for (int r = 0; r < P; ++r) {
    if (rdlane(order_brush, r) == lane) rank_brush = r;
    if (rdlane(order_claim, r) == lane) rank_claim = r;
  }
  bool fire_zap = false, fire_claim = false, mark_reset = false;
  if (is_av) {
    // 140 Zapper zap (avatar_library.lua:613-636)
    if (a.alive && c.zap.cooldown >= 0) {
      if (a.ztimer > 0) a.ztimer--;
      else if (a_zap == 1) { a.ztimer = c.zap.cooldown; fire_zap = true; }
    }
    // 100 ResourceClaimer claim (territory/components.lua:255-275)
    if (c.claim_wait >= 0) {
      if (a.ctimer > 0) a.ctimer--;
      else if (a_claim == 1) { a.ctimer = c.claim_wait; fire_claim = true; }
    }
  }
  const unsigned long long alive_pre = __ballot(is_av && a.alive);
  // 100 StochasticIntervalEpisodeEnding: _t == step + 1
  int cont = is_reset ? 1 : tail->cont;
  if (frame >= c.ee_min_frames && (step + 1) % c.ee_interval == 0)
    if (philox_u53(draw(RS_EPISODE_END, 0)) < c.thr_ee) cont = 0;
  #pragma unroll
  for (int k = 0; k < kResPerLane; ++k) {
    const int cell = rcell[k], i = k * 64 + lane;
    if (cell < 0) continue;
    const int rs = at(c.res_layer, cell);
    // group claimedResources only (an avatar may stand on a destroyed resource's
    // cell: the resource shares the avatars' layer)
    if (ts->owner[rs] < 0) continue;
    int A = at(c.plane_a, cell);
    const int age = at(c.plane_c, cell), cb = A >> 3;
    // 100 Resource provideRewards: probability rewardRate, startFrame rewardDelay
    // (territory/components.lua:85-102)
    if (age >= c.reward_delay && philox_u53(draw(RS_RESOURCE_REWARD, (uint32_t)i)) < c.thr_reward) {
      if (cb > 0) atomicAdd(&ts->reward_count[cb - 1], 1);
      A |= 4;
    }
    // 2 Resource releaseClaimOfDeadAgent: startFrame 5 (:103-117)
    if (age >= 5 && cb > 0 && !((alive_pre >> (cb - 1)) & 1ull)) {
      mark[cell] |= 1;  // setState(unclaimed), applied at the end of flush 1
      A &= 3;           // _rewardingStatus = inactive, _claimedByAvatarComponent = nil
    }
    at(c.plane_a, cell) = (uint8_t)A;
  }
  wsync();
  if (is_av) {
    // Avatar:addReward of provideRewards (Taste role 'none'), skipped in wait state
    const int cnt = ts->reward_count[lane];
    if (a.alive) for (int k = 0; k < cnt; ++k) a.reward += c.reward;
    // 3 GraduatedSanctionsMarking resetToInitialLevel (avatar_library.lua:1010-1026)
    if (level != 1 && a.alive) {
      tsince++;
      if (tsince == c.recovery_time) {
        level = 1; mark_reset = true; tsince = 0;
        push_event(sc, MP_EVENT_SET_SANCTIONING_LEVEL, lane + 1, 1);
      }
    }
  }

  TSTAMP(3);
  // ---- flush 1, FIFO
  // Avatar scheduled removal: setState(wait) queued by Avatar:update; 'die'
  // sends the marking to its wait state in the next flush.
  bool died = false;
  if (is_av && remove_now && a.alive) {
    at(t.avatar_layer, a.y * W + a.x) = 0;
    a.alive = 0; a.achange = frame; died = true;
  }
  // Avatar move (avatar_library.lua:155-203): turn (self + connected), moveRel;
  // the connected marking moves with the avatar.
  resolve_moves(t, wd, a, mov_allowed ? a_move : 0, mov_allowed ? a_turn : 0, order_move,
                alive_state, c.mark_layer);
  const int mstate_before = mstate;  // what the plane holds at the marking's cell
  if (lane < MP_MAX_PLAYERS) {
    ts->av_ori[lane] = (uint8_t)a.ori;
    ts->mark_cell[lane] = (int16_t)((is_av && mstate > 0) ? a.y * W + a.x : -1);
  }
  wsync();

  TSTAMP(4);
  const BeamLane zap_lane = beam_lane(c.zap.shape, lane);
  // zapHit beams one at a time, in visiting order (Resource:onHit changes _health
  // immediately, territory/components.lua:155-181)
  int mark_level_pending = 0;
  const int zap_firing = (int)(fire_zap && a.alive);
  for (int r = 0; r < P; ++r) {
    const int b = rdlane(order_zap, r);
    if (rdlane(zap_firing, b) == 0) continue;
    fire_beams(t, wd, tail, a, fire_zap, zap_lane, c.zap.hit, false,
               c.zap.layer, c.zap.s_hit, false,
               [&](int s, int cell) {
                 if ((wd.sinfo[s] >> 24) != 0) return 1;  // Zapper:onHit stops the zap
                 if (s == c.s_mark[0] || s == c.s_mark[1]) {
                   // GraduatedSanctionsMarking:onHit: report it for the marking's
                   // avatar; the marking itself does not stop the beam
                   for (int p = 0; p < P; ++p)
                     if (ts->mark_cell[p] == cell) return ((p + 1) << 8);
                   return 0;
                 }
                 if (s != c.s_res_unclaimed && ts->owner[s] < 0) return 0;
                 // a resource stops the zap unless this hit destroys it
                 return ((at(c.plane_a, cell) & 3) - 1 != 0) ? 3 : 2;
               },
               [&](int, int, int, bool reached, int cell, bool touched) {
                 if (reached) {  // Zapper:onHit of an avatar standing there
                   const int pl = (int)(wd.sinfo[at(t.avatar_layer, cell)] >> 24) - 1;
                   if (pl >= 0) push_event(sc, MP_EVENT_ZAP, b + 1, pl + 1);
                 }
                 if (!touched) return;
                 int A = at(c.plane_a, cell);
                 int health = (A & 3) - 1;
                 at(c.plane_b, cell) = 1;  // _framesSinceZapped = 0
                 if (health == 0) {
                   health = c.initial_health;
                   A &= ~4;                // _rewardingStatus = inactive
                   mark[cell] |= 2;        // destroyed: state changes in the next flush
                   push_event(sc, MP_EVENT_DESTROYED_RESOURCE, b + 1, 0);
                 }
                 at(c.plane_a, cell) = (uint8_t)((A & ~3) | health);
               },
               b);
    // GraduatedSanctionsMarking:onHit for every avatar this beam reached
    // (avatar_library.lua:1051-1097); the marking shares the avatar's cell
    for (int j = 0; j < c.zap.shape.n; ++j) {
      const int v = __builtin_amdgcn_readfirstlane((int)sc->victim[b][j]);
      if (v < 0) continue;
      const int l = rdlane(level, v) - 1;   // (levels are 1 or 2: selects, not indexed loads)
      const double lv_source = l ? c.lv_source[1] : c.lv_source[0];
      const double lv_target = l ? c.lv_target[1] : c.lv_target[0];
      const int lv_increment = l ? c.lv_increment[1] : c.lv_increment[0];
      const int lv_remove = l ? c.lv_remove[1] : c.lv_remove[0];
      const int lv_freeze = l ? c.lv_freeze[1] : c.lv_freeze[0];
      if (lane == b && a.alive) a.reward += lv_source;
      if (lane == v) {
        if (a.alive) a.reward += lv_target;
        level += lv_increment;
        push_event(sc, MP_EVENT_SANCTIONING, b + 1, v + 1);
        if (lv_remove) {
          removal = 1; mov_allowed = 0; freeze = 1; disallow = 1; nozap = 1;
          push_event(sc, MP_EVENT_REMOVAL_DUE_TO_SANCTIONING, b + 1, v + 1);
        } else {
          mark_level_pending = level;  // _setLevel, next flush
          push_event(sc, MP_EVENT_SET_SANCTIONING_LEVEL, v + 1, level);
          if (lv_freeze > 0) {
            mov_allowed = 0; freeze = lv_freeze; disallow = 1; nozap = lv_freeze;
          }
        }
        tsince = 0;
      }
    }
    wsync();
  }

  TSTAMP(5);
  // 130 Paintbrush (directionHit<i>, length 1, every frame, every on-grid avatar)
  // and 100 ResourceClaimer (claimBeam_<i>, radius 0; passes resources and
  // avatars, stopped by AllBeamBlocker walls only).  At most P + P * len beam
  // cells exist per frame, so they are kept in lanes — entry = tag | claimable
  // << 14 | by-non-owner << 15 | cell << 16, tag = (visit rank << 8 | player) + 1
  // with claim beams ranked after all brushes — and "the last beam over a cell
  // wins" (events are processed in order) is a compare loop over the entries.
  constexpr uint32_t kNoEntry = 0xffff0000u;
  uint32_t eb = kNoEntry, ec = kNoEntry;
  if (is_av && a.alive) {
    int x = a.x, y = a.y;
    if (step_cell(t, x, y, dir_dx(a.ori), dir_dy(a.ori))) {
      const int cell = y * W + x;
      const uint32_t tag = ((uint32_t)rank_brush << 8 | (uint32_t)lane) + 1u;
      const int rs = at(c.res_layer, cell);
      const bool claimable = rs == c.s_res_unclaimed || ts->owner[rs] >= 0;  // Resource:_claim
      const bool differs = claimable && rs != ts->s_claimed[lane];
      eb = tag | ((uint32_t)claimable << 14) | ((uint32_t)differs << 15) | ((uint32_t)cell << 16);
    }
  }
  const int len = c.claim_length;
  const int cp = lane / len, cf = lane - cp * len + 1;   // claim lane = (player, ray cell)
  const int cps = cp < P ? cp : 0;
  {
    const bool lane_ok = cp < P;
    const bool fire = __shfl((int)(fire_claim && a.alive), cps) != 0 && lane_ok;
    const int px = __shfl(a.x, cps), py = __shfl(a.y, cps), po = __shfl(a.ori, cps);
    const int prank = __shfl(rank_claim, cps);
    int x = px, y = py;
    const bool inb = step_cell(t, x, y, cf * dir_dx(po), cf * dir_dy(po));
    const int cell = inb ? y * W + x : 0;
    bool blocked = false;
    if (fire && inb) {
      const uint32_t hbit = ts->hit_claim[cps];
      for (int l0 = 0; l0 < t.L; l0 += 4) {   // four planes per LDS round trip (fire_beams)
        uint32_t st[4], info[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) st[k] = l0 + k < t.L ? at(l0 + k, cell) : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) info[k] = wd.sinfo[st[k]];   // sinfo[0] == 0
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (st[k] != 0 && ((info[k] >> hbit) & 1u)) blocked = true;
      }
    }
    const unsigned long long stops = __ballot(fire && (!inb || blocked));
    const uint32_t mine = (uint32_t)(stops >> (cp * len)) & ((1u << len) - 1u);
    const bool reached = fire && inb && (mine & ((1u << (cf - 1)) - 1u)) == 0;
    if (reached) {   // A4: drawn on the blocked cell too
      const uint32_t tag = (((uint32_t)(16 + prank)) << 8 | (uint32_t)cps) + 1u;
      const int rs = at(c.res_layer, cell);
      const bool claimable = rs == c.s_res_unclaimed || ts->owner[rs] >= 0;
      const bool differs = claimable && rs != ts->s_claimed[cps];
      ec = tag | ((uint32_t)claimable << 14) | ((uint32_t)differs << 15) | ((uint32_t)cell << 16);
    }
  }
  // Resource:_claim by someone who is not the owner yet reports the claim
  // (not on a resource a zap of this flush has just destroyed: _destroyed is set at once)
  if (eb != kNoEntry && (eb & 0x8000u) && !(mark[eb >> 16] & 2))
    push_event(sc, MP_EVENT_CLAIMED_RESOURCE, lane + 1, 0);
  if (ec != kNoEntry && (ec & 0x8000u) && !(mark[ec >> 16] & 2))
    push_event(sc, MP_EVENT_CLAIMED_RESOURCE, cps + 1, 0);
  // per entry: is it the last of its kind / the last claimable / the last
  // by-non-owner on its cell?
  bool b_top = eb != kNoEntry, b_call = (eb >> 14) & 1u, b_diff = (eb >> 15) & 1u;
  bool c_top = ec != kNoEntry, c_call = (ec >> 14) & 1u, c_diff = (ec >> 15) & 1u;
  if (eb == kNoEntry) { b_call = false; b_diff = false; }
  if (ec == kNoEntry) { c_call = false; c_diff = false; }
  {
    const int n_src = P * len > P ? P * len : P;
    const uint32_t btag = eb & 0x3fffu, ctag = ec & 0x3fffu;
    for (int q = 0; q < n_src; ++q) {
      const uint32_t qb = (uint32_t)rdlane((int)eb, q), qc = (uint32_t)rdlane((int)ec, q);
      const uint32_t qbt = qb & 0x3fffu, qct = qc & 0x3fffu;
      if ((qb >> 16) == (eb >> 16) && qbt > btag) {
        b_top = false;
        if (qb & 0x4000u) b_call = false;
        if (qb & 0x8000u) b_diff = false;
      }
      if ((qc >> 16) == (eb >> 16) && qct > btag) {
        if (qc & 0x4000u) b_call = false;
        if (qc & 0x8000u) b_diff = false;
      }
      if ((qc >> 16) == (ec >> 16) && qct > ctag) {
        c_top = false;
        if (qc & 0x4000u) c_call = false;
        if (qc & 0x8000u) c_diff = false;
      }
      if ((qb >> 16) == (ec >> 16) && qbt > ctag) {
        if (qb & 0x4000u) c_call = false;
        if (qb & 0x8000u) c_diff = false;
      }
    }
  }
  wsync();
  // beam sprites + the _claim bookkeeping of the cell, written by the winners
  if (b_top) at(c.brush_layer, eb >> 16) = ts->s_brush[lane][a.ori & 3];
  if (c_top) at(c.claim_layer, ec >> 16) = ts->s_claim_hit[cps];
  if (b_call) lastcall[eb >> 16] = (uint16_t)(eb & 0x3fffu);
  if (c_call) lastcall[ec >> 16] = (uint16_t)(ec & 0x3fffu);
  if (b_diff) lastdiff[eb >> 16] = (uint16_t)(eb & 0x3fffu);
  if (c_diff) lastdiff[ec >> 16] = (uint16_t)(ec & 0x3fffu);
  wsync();
  // end of flush 1: the resetToInitialLevel _setLevel and the released claims
  if (is_av && mark_reset && mstate > 0) mstate = 1;
  #pragma unroll
  for (int k = 0; k < kResPerLane; ++k) {
    const int cell = rcell[k], i = k * 64 + lane;
    if (cell < 0) continue;
    // Resource:_claim bookkeeping of this flush: the last caller owns
    // _claimedByAvatarComponent; a caller who is not the current owner (and finds
    // the resource not destroyed) queues setState and clears the reward status
    const uint32_t lc = lastcall[cell], ld = lastdiff[cell];
    int A = at(c.plane_a, cell);
    if (lc) A = (A & 7) | ((int)(((lc - 1u) & 255u) + 1u) << 3);
    const bool destroyed_now = (mark[cell] & 2) != 0;
    if (ld && !destroyed_now) A &= ~4;
    at(c.plane_a, cell) = (uint8_t)A;
    if ((mark[cell] & 1) && ts->owner[at(c.res_layer, cell)] >= 0) {
      at(c.res_layer, cell) = (uint8_t)c.s_res_unclaimed;
      at(c.plane_c, cell) = 0;
    }
  }
  wsync();

  TSTAMP(6);
  // ---- flush 2: setStates queued by the callbacks of flush 1
  // (marking: 'die' queued its wait state at the start of flush 1, a zap's
  // _setLevel was queued later — the level lands last)
  if (died) mstate = 0;
  if (is_av && mark_level_pending > 0) mstate = mark_level_pending;
  #pragma unroll
  for (int k = 0; k < kResPerLane; ++k) {
    const int cell = rcell[k], i = k * 64 + lane;
    if (cell < 0) continue;
    if (mark[cell] & 2) {  // destroyed resource + its texture + its damage indicator
      at(c.res_layer, cell) = 0;
      at(c.tex_layer, cell) = 0;
      at(c.dmg_layer, cell) = (uint8_t)c.s_dmg_inactive;
    } else if (lastdiff[cell] && (at(c.res_layer, cell) == c.s_res_unclaimed ||
                                  ts->owner[at(c.res_layer, cell)] >= 0)) {
      const int ns = ts->s_claimed[((uint32_t)lastdiff[cell] - 1u) & 255u];
      if (at(c.res_layer, cell) != ns) {
        at(c.res_layer, cell) = (uint8_t)ns;
        at(c.plane_c, cell) = 0;
      }
    }
    mark[cell] = 0;
    // one frame older (grid:frames)
    const int age = at(c.plane_c, cell);
    if (age < 255) at(c.plane_c, cell) = (uint8_t)(age + 1);
  }
  wsync();
  // (a marking long gone must not touch its old cell: someone else may stand there)
  if (is_av && mstate != mstate_before)
    at(c.mark_layer, a.y * W + a.x) = (uint8_t)(mstate ? c.s_mark[mstate - 1] : 0);
  int claimed = 0;
#pragma unroll
  for (int k = 0; k < kResPerLane; ++k) {
    const int rs = rcell[k] >= 0 ? at(c.res_layer, rcell[k]) : 0;
    claimed += __popcll(__ballot(ts->owner[rs] >= 0));
  }
  const unsigned long long badb = __ballot(bad != 0);
  const int done = is_reset ? 0 : !(cont && step < t.max_frames);
  if (lane == 0) {
    tail->step = step;
    tail->frame = frame + 1;
    tail->cont = cont;
    tail->done = done;
    tail->aux_count = claimed;
    if (!is_reset) { tail->ctr[0]++; tail->ctr[1] += (uint32_t)P; tail->ctr[7] += __popcll(badb); }
  }
  if (lane < MP_MAX_PLAYERS) {
    tail->flag0[lane] = (uint8_t)mstate;
    tail->freeze[lane] = (uint8_t)freeze; tail->removal[lane] = (uint8_t)removal;
    tail->aflags[lane] = (uint8_t)(mov_allowed | (disallow << 1));
    tail->nozap[lane] = (uint8_t)nozap; tail->level[lane] = (uint8_t)level;
    tail->tsince[lane] = (uint8_t)tsince;
  }
  const int step_type = is_reset ? 0 : (done ? 2 : 1);
  TSTAMP(7);
  finish(t, wd, tail, a, 0.0, c.zap.cooldown, step_type, out, kOrders);
  TSTAMP(8);
#ifdef MP_STEP_TIMING
  if (lane == 0 && (w == 7 || w == 5000) && !is_reset)
    printf("w %d: load %llu simupd %llu updaters %llu moves %llu zaps %llu brushclaim %llu flush2 %llu finish %llu total %llu\n",
           w, ts_[1] - ts_[0], ts_[2] - ts_[1], ts_[3] - ts_[2], ts_[4] - ts_[3], ts_[5] - ts_[4],
           ts_[6] - ts_[5], ts_[7] - ts_[6], ts_[8] - ts_[7], ts_[8] - ts_[0]);
#endif
}

}  // namespace stepk

#endif  // MP_STEP_TERRITORY_H_
