// step_mushroom.h — one environment step (or episode start) of one externality_mushrooms
// world by one wavefront (shape: step_territory.h / step_common.h).
//
// Substrate rules restated here (reference: configs/substrates/externality_mushrooms.py,
// externality_mushrooms__dense.py; lua/levels/externality_mushrooms/components.lua;
// lua/modules/avatar_library.lua):
//   MushroomEating     components.lua:30-153   who a mushroom of each type pays, its spores,
//                                              what it destroys, how long it is digested
//   MushroomGrowable / MushroomRegrowth  :155-255   the potential sites and their counter
//   Destroyable        :257-306                a zap destroys a mushroom (health 1) and passes
//   Perishable         :308-335                a mushroom lives `delay` frames of its type
//   GraduatedSanctionsMarking  avatar_library.lua:948-1121   as territory, plus its 'respawn'
//   Avatar / Zapper    timed freeze, zap prevention, scheduled removal, respawn after 50 frames
//
// The potential sites.  The Lua keeps a set of pieces and a counter, both brought up to date
// by a priority-500 updater from flags that onStateChange left in earlier frames.  With A20
// (oracle/externality_mushrooms.c: the pieces' creation raises those flags too) the set a
// frame draws from is exactly the mushrooms that were in their wait state when the frame
// began, and the counter runs below its size by the number of mushrooms the map starts with:
// a site is never both grown and sent to wait in one frame (a grow needs it waiting at the
// frame's start, every road to waiting needs it live then).  So the kernel keeps neither:
// the ballots of "waiting at the frame's start" ARE the sorted set.
// Per cell: the mushroom's state on the lowerPhysical plane, and one hidden plane behind the
// render planes with the frames since its last state change, saturating (grid:frames(piece)
// for Perishable's startFrame).  What flush 2 does to a site is collected in the scratch
// marks: bit 0 = it waits, bits 1-3 = 1 + the type it grows into (the last grow wins).
//
// Eating is sequential where the Lua is: one eater at a time, in the order the frame's moves
// (then its respawns) are processed, each spore's draw looking for avatars where they stand
// at THAT point of the flush (components.lua:224-231).
//
// The marking (GraduatedSanctionsMarking's piece on the superOverlay layer), literally: every
// avatar's marking has a state (tail->flag0: 0 = its wait state, off the grid; 1, 2 = level_k,
// a byte on the marking plane) and a POSITION OF ITS OWN (tail->ctimer / flag1: x, y; a piece
// keeps its transform while off the grid, A18) — usually the avatar's cell, but not always,
// since avatars come back here (avatar_library.lua:1099-1110):
//   * 'die' sends it to wait where it is; a zap's _setLevel queued behind that brings it back
//     there — an orphan, on the map without its avatar;
//   * 'respawn': _setLevel puts it back WHERE IT WAITED — unless another marking lies there:
//     it then stays in its wait state (its avatar can no longer be sanctioned) — and the
//     teleport takes it to its avatar — unless another marking (an orphan) lies on the spawn
//     cell: it then stays where it is, CONNECTED AT A DISTANCE, and from then on the two move
//     as one group, each needing its own target free (A14); the teleport of a marking that is
//     off the grid only moves its transform;
//   * resetToInitialLevel's _setLevel brings a marking that never came back onto the map at
//     its transform — where its avatar respawned, however far that avatar has walked since.
// Round 5 kept the marking at its avatar's cell and COUNTED the respawns and resets that
// leave it elsewhere (MP_CTR_AUX0); round 6 restates them: moves are resolved one avatar at a
// time in visiting order against the planes as they are (five avatars: cheaper than it
// sounds), everything else reads the marking's own position.  MP_CTR_AUX0 still counts how
// often a marking ends a frame on the map away from its living avatar (a statistic: the tests
// that reach these cases ask for it to be non-zero).
#ifndef MP_STEP_MUSHROOM_H_
#define MP_STEP_MUSHROOM_H_

#include "step_common.h"

namespace stepk {

constexpr int kShroomPerLane = 4;   // mp_create admits at most 64 * kShroomPerLane sites

struct MushroomSites { int site[kShroomPerLane]; };   // this lane's sites (i = k * 64 + lane)

__device__ inline MushroomSites load_sites(const MushroomTables& c, int lane) {
  MushroomSites s;
#pragma unroll
  for (int k = 0; k < kShroomPerLane; ++k)
    s.site[k] = k * 64 + lane < c.n_site ? c.site_cells[k * 64 + lane] : -1;
  return s;
}

// LDS behind the marks of a scratch slot.
struct EmScratch { int16_t mark_cell[MP_MAX_PLAYERS]; };   // cell of avatar p's marking overlay, or -1
static_assert(sizeof(EmScratch) % 16 == 0, "EmScratch");
__host__ __device__ inline int extra_bytes(const MushroomTables&) { return (int)sizeof(EmScratch); }

// spawn_avatars (step_common.h) for spawn groups of up to 256 cells: the "dense" map makes
// every free cell a spawn point.  No 'choice' spawn points (mp_create refuses them here).
__device__ inline void spawn_avatars_wide(const DevTables& t, uint8_t* grid, int lane, uint32_t ep,
                                          uint32_t k0, uint32_t k1, Av& a) {
  const int P = t.P, HW = t.H * t.W;
  const bool is_av = lane < P;
  const int my_group = is_av ? t.avatar_init_group[lane] : -1;
  int my_cell = 0;
  for (int g = 0; g < t.n_init_groups; ++g) {
    const int base = t.init_spawn_ptr[g];
    const int ns = t.init_spawn_ptr[g + 1] - base;
    const unsigned long long members = __ballot(my_group == g);
    const int want = __popcll(members);
    // the pool: position q lives in lane q & 63, register q >> 6 (named scalars and uniform
    // branches: nothing here may become an indexed register access)
    int i0 = lane < ns ? t.init_spawn_cells[base + lane] : 0;
    int i1 = 64 + lane < ns ? t.init_spawn_cells[base + 64 + lane] : 0;
    int i2 = 128 + lane < ns ? t.init_spawn_cells[base + 128 + lane] : 0;
    int i3 = 192 + lane < ns ? t.init_spawn_cells[base + 192 + lane] : 0;
    int j = lane;
    if (lane < want && lane < ns)
      j = lane + (int)philox_bounded(
          philox4x32_10((uint32_t)(lane + 256 * g), RS_START_SPAWN, 0u, ep, k0, k1),
          (uint32_t)(ns - lane));
    for (int i = 0; i < want; ++i) {   // (want <= 16: position i lives in i0 of lane i)
      const int ji = __builtin_amdgcn_readfirstlane(__shfl(j, i));
      if (ji == i) continue;
      const int r = ji >> 6, jl = ji & 63;
      const int vi = __shfl(i0, i);
      int vj;
      if (r == 0) vj = __shfl(i0, jl); else if (r == 1) vj = __shfl(i1, jl);
      else if (r == 2) vj = __shfl(i2, jl); else vj = __shfl(i3, jl);
      // (position ji first: it may be lane i's own lane in another register)
      if (r == 0) { if (lane == jl) i0 = vi; }
      else if (r == 1) { if (lane == jl) i1 = vi; }
      else if (r == 2) { if (lane == jl) i2 = vi; }
      else { if (lane == jl) i3 = vi; }
      if (lane == i) i0 = vj;
    }
    const int rank = __popcll(members & ((1ull << lane) - 1ull));
    const int got = __shfl(i0, my_group == g ? rank : 0);
    if (my_group == g) my_cell = got;
  }
  a = Av();
  if (is_av) {
    a.ori = (int)philox_bounded(philox4x32_10((uint32_t)lane, RS_START_ORIENT, 0u, ep, k0, k1), 4u);
    a.x = my_cell % t.W; a.y = my_cell / t.W; a.alive = 1;
    grid[t.avatar_layer * HW + my_cell] = (uint8_t)t.alive_state[lane];
  }
}

__device__ inline void step_world(const DevTables& t, const MushroomTables& c,
                                  const MushroomSites& sites, const World& wd,
                                  const Action& act, const StepArgs& args) {
  const int lane = wd.lane, w = wd.w;
  const StepOutputs& out = args.out;
  Scratch* sc = wd.sc;
  const int P = t.P, HW = t.H * t.W, W = t.W;
  uint8_t* mark = wd.mark;   // bit 0: waits from flush 2 on; bits 1-3: 1 + the type it grows into
  EmScratch* es = reinterpret_cast<EmScratch*>(wd.extra);
  uint8_t* grid = wd.rec;
  WorldTail* tail = reinterpret_cast<WorldTail*>(wd.rec + t.grid_pad);
  const bool is_av = lane < P;
  auto at = [&](int layer, int cell) -> uint8_t& { return grid[layer * HW + cell]; };

  const OrderStreams kOrders = {RS_SHUFFLE_MOVE, RS_SHUFFLE_ZAP, RS_SHUFFLE_RESPAWN, 0, 3};   // the updater groups shuffled per frame (A1)

  const int what = dispatch(t, tail, lane, w, args.reset_mask, args.mode, args.auto_reset, out);
  if (what == 0) return;
  const bool is_reset = what == 1;
  const int alive_state = is_av ? t.alive_state[lane] : 0;

  Av a;
  int freeze = 0, removal = 0, mov_allowed = 1, disallow = 0, nozap = 0, level = 1, tsince = 0;
  int mstate = 0;  // marking piece: 0 wait (off-grid), 1 level_1, 2 level_2
  int mx = 0, my = 0;   // ... and its transform
  int a_move = 0, a_turn = 0, a_zap = 0, bad = 0;
  bool remove_now = false;
  uint32_t k0, k1, ep;
  int step, frame;

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
    for (int i = lane; i < HW; i += 64) at(c.plane_age, i) = 0;   // (and the pad bytes copied above)
    if (lane == 0) {
      tail->episode = ep + 1;
      tail->done = 0; tail->cont = 1; tail->started = 1;
      tail->group_change = 0;
      tail->ctr[2]++;
    }
    if (lane < MP_MAX_PLAYERS) { tail->flag0[lane] = 0; tail->flag1[lane] = 0; }
    wsync();
    spawn_avatars_wide(t, grid, lane, ep, k0, k1, a);
    // GraduatedSanctionsMarking:postStart (avatar_library.lua:1034-1049)
    if (is_av) {
      mstate = 1; mx = a.x; my = a.y; at(c.mark_layer, a.y * W + a.x) = (uint8_t)c.s_mark[0];
      push_event(sc, MP_EVENT_AVATAR_STARTED, 0, 0);
      push_event(sc, MP_EVENT_SET_SANCTIONING_LEVEL, lane + 1, 1);
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
      mx = a.ctimer; my = tail->flag1[lane];   // (this level has no other use for either)
    }
    a_move = act.move; a_turn = act.turn; a_zap = act.fire0; bad = act.bad;
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
  }
  auto draw = [&](int stream, uint32_t index) {
    return philox4x32_10(index, (uint32_t)stream, (uint32_t)step, ep, k0, k1);
  };
  // beam sprites of the previous frame disappear (grid:update start)
  clear_bytes(grid, c.zap.layer * HW, HW, lane);
  wsync();

  // ---- the sites as the frame finds them.  500 MushroomGrowable registration: the waiting
  // ones are this frame's potential sites; 3 Perishable: the old ones wait from the end of
  // flush 1 on
  int stype[kShroomPerLane];
  unsigned long long wait_m[kShroomPerLane];
  uint32_t perish_bits = 0;
#pragma unroll
  for (int k = 0; k < kShroomPerLane; ++k) {
    const int cell = sites.site[k];
    const int s = cell >= 0 ? at(c.live_layer, cell) : 255;
    wait_m[k] = __ballot(s == 0);
    const int ty = (cell >= 0 && s != 0) ? s - c.s_type0 : -1;
    // (a byte per type, 255 = never: a per-lane index into a kernel argument would cost a
    // scratch copy of the whole struct)
    const int delay = (int)((c.perish_packed >> (8 * (ty & 3))) & 255u);
    if (ty >= 0 && delay != 255 && at(c.plane_age, cell) >= delay) perish_bits |= 1u << k;
    stype[k] = ty;
  }
  const int n_wait = __popcll(wait_m[0]) + __popcll(wait_m[1]) + __popcll(wait_m[2]) + __popcll(wait_m[3]);
  const int potential = n_wait - c.n_live_init;   // MushroomRegrowth._numPotentialMushrooms

  // ---- updaters, priority descending; they read the pre-flush state
  int orders[4];
  step_orders(tail, lane, P, kOrders, (uint32_t)step, ep, k0, k1, orders);
  const int order_move = orders[0], order_zap = orders[1], order_resp = orders[2];
  int rank_move = 0, rank_resp = 0;  // inverse permutations
  for (int r = 0; r < P; ++r) {
    if (rdlane(order_move, r) == lane) rank_move = r;
    if (rdlane(order_resp, r) == lane) rank_resp = r;
  }
  bool fire_zap = false, mark_reset = false, want_respawn = false;
  if (is_av) {
    // 140 Zapper zap (avatar_library.lua:613-636)
    if (a.alive && c.zap.cooldown >= 0) {
      if (a.ztimer > 0) a.ztimer--;
      else if (a_zap == 1) { a.ztimer = c.zap.cooldown; fire_zap = true; }
    }
    // 135 Zapper respawn: state = waitState, startFrame = framesTillRespawn (:638-649)
    want_respawn = !a.alive && (frame - a.achange) >= c.zap.respawn_frames;
  }
  // 100 StochasticIntervalEpisodeEnding: _t == step + 1
  int cont = is_reset ? 1 : tail->cont;
  if (frame >= c.ee_min_frames && (step + 1) % c.ee_interval == 0)
    if (philox_u53(draw(RS_EPISODE_END, 0)) < c.thr_ee) cont = 0;
  if (is_av) {
    // 3 GraduatedSanctionsMarking resetToInitialLevel (avatar_library.lua:1010-1026)
    if (level != 1 && a.alive) {
      tsince++;
      if (tsince == c.recovery_time) {
        level = 1; mark_reset = true; tsince = 0;
        push_event(sc, MP_EVENT_SET_SANCTIONING_LEVEL, lane + 1, 1);
      }
    }
  }

  // ---- flush 1, FIFO
  // Avatar scheduled removal: setState(wait) queued by Avatar:update; 'die' sends the marking
  // to its wait state in the next flush.
  bool died = false;
  if (is_av && remove_now && a.alive) {
    at(t.avatar_layer, a.y * W + a.x) = 0;
    a.alive = 0; a.achange = frame; died = true;
  }
  const int old_cell = a.y * W + a.x;
  // Avatar move (avatar_library.lua:155-203): turn (self + connected), moveRel — do_move
  // (oracle/engine.c) for the group {avatar, its marking if that is on the map}, one avatar at
  // a time in visiting order against the planes as they are: the move succeeds only if EVERY
  // member's target is free (A14), and the marking may be somewhere else than its avatar.
  const int mv = mov_allowed ? a_move : 0, tn = mov_allowed ? a_turn : 0;
  if (is_av && tn != 0) a.ori = (a.ori + tn + 4) & 3;   // off-grid pieces turn too
  const bool wants = is_av && a.alive && mv != 0;
  wsync();
  if (__ballot(wants) != 0ull)
    for (int r = 0; r < P; ++r) {
      const int p = rdlane(order_move, r);
      if (lane == p && wants) {
        const int dir = (a.ori + mv - 1) & 3;
        int tx = a.x, ty = a.y, fx = mx, fy = my;
        const bool grouped = mstate > 0;
        bool ok = step_cell(t, tx, ty, dir_dx(dir), dir_dy(dir)) && at(t.avatar_layer, ty * W + tx) == 0;
        if (ok && grouped)
          ok = step_cell(t, fx, fy, dir_dx(dir), dir_dy(dir)) && at(c.mark_layer, fy * W + fx) == 0;
        if (ok) {
          at(t.avatar_layer, a.y * W + a.x) = 0;
          at(t.avatar_layer, ty * W + tx) = (uint8_t)alive_state;
          a.x = tx; a.y = ty;
          if (grouped) {
            const uint8_t m = at(c.mark_layer, my * W + mx);
            at(c.mark_layer, my * W + mx) = 0;
            at(c.mark_layer, fy * W + fx) = m;
            mx = fx; my = fy;
          }
        }
      }
      wsync();
    }
  const int new_cell = a.y * W + a.x;   // (an avatar that is away keeps the cell it left from)
  if (lane < MP_MAX_PLAYERS)
    es->mark_cell[lane] = (int16_t)((is_av && mstate > 0) ? my * W + mx : -1);
  wsync();

  // MushroomEating:onEnter (components.lua:107-138) of eater p on a type-T mushroom at `mcell`,
  // at rank `rnk` of the moves (or of the respawns, which come after every move and beam)
  bool respawned = false;
  // (inlined at both call sites: a closure called out of line keeps what it captures in scratch memory)
  auto eat = [&](int p, int T, int mcell, bool resp_phase, int rnk) __attribute__((always_inline)) {
    if (lane == p) push_event(sc, MP_EVENT_EATING_MUSHROOM, p + 1, T + 1);
    // who stands where at this point of the flush
    const bool here_now = is_av && a.alive && (!resp_phase || !respawned || rank_resp <= rnk);
    const int cell_now = (resp_phase || rank_move <= rnk) ? a.y * W + a.x : old_cell;
    // _rewardEveryone (:65-105); Avatar:addReward skips an avatar in its wait state
    const double rs = T == 0 ? c.rew_self[0] : T == 1 ? c.rew_self[1] : T == 2 ? c.rew_self[2] : c.rew_self[3];
    const double ro = T == 0 ? c.rew_other[0] : T == 1 ? c.rew_other[1] : T == 2 ? c.rew_other[2] : c.rew_other[3];
    if (here_now) {
      if (lane == p) { if ((c.pays >> T) & 1u) a.reward += rs; }
      else if ((c.pays >> (4 + T)) & 1u) a.reward += ro;
    }
    const int spores = c.i32[8 + T], digest = c.i32[12 + T], destroy = c.i32[20 + T];
    // MushroomRegrowth:grow per spore (:216-235)
    for (int n = 0; n < spores; ++n) {
      for (int m = 0; m < 4; ++m) {
        if (potential < c.min_potential || n_wait == 0) continue;
        const Philox4 d = draw(RS_MUSHROOM_GROW, (uint32_t)((p * 4 + n) * 4 + m));
        if (philox_u53(d) >= c.thr[T * 4 + m]) continue;
        const int site = __builtin_amdgcn_readfirstlane(
            kth_site(wait_m[0], wait_m[1], wait_m[2], wait_m[3], (int)philox_bounded(d, (uint32_t)n_wait)));
        const int r = site >> 6;
        const int cell = rdlane(r == 0 ? sites.site[0] : r == 1 ? sites.site[1] : r == 2 ? sites.site[2]
                                                                                            : sites.site[3],
                                site & 63);
        if (__ballot(here_now && cell_now == cell) != 0ull) continue;   // queryPosition('upperPhysical')
        if (lane == 0) mark[cell] = (uint8_t)((mark[cell] & 1) | ((m + 1) << 1));
      }
    }
    // destroyRandomMushrooms (:237-244): every mushroom of the type, each with the probability
    if (destroy >= 0) {
#pragma unroll
      for (int k = 0; k < kShroomPerLane; ++k)
        if (stype[k] == destroy &&
            philox_u53(draw(RS_MUSHROOM_DESTROY, (uint32_t)(p * 256 + k * 64 + lane))) < c.thr[16 + T])
          mark[sites.site[k]] |= 1;
    }
    if (lane == p && digest > 0) { mov_allowed = 0; freeze = digest; }   // disallowMovementUntil
    if (lane == 0) mark[mcell] |= 1;   // the eaten mushroom waits from flush 2 on
    wsync();
  };

  {
    int eat_ty = -1;   // A3b: a blocked move enters in place
    if (wants) { const int s = at(c.live_layer, new_cell); if (s != 0) eat_ty = s - c.s_type0; }
    if (__ballot(eat_ty >= 0) != 0ull)
      for (int r = 0; r < P; ++r) {
        const int p = rdlane(order_move, r);
        const int T = rdlane(eat_ty, p);
        if (T >= 0) eat(p, T, rdlane(new_cell, p), false, r);
      }
  }

  const BeamLane zap_lane = beam_lane(c.zap.shape, lane);
  // zapHit beams one at a time, in visiting order (GraduatedSanctionsMarking:onHit reads the
  // level the previous beam left, avatar_library.lua:1051-1097)
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
                   for (int p = 0; p < P; ++p)
                     if (es->mark_cell[p] == cell) return ((p + 1) << 8);
                   return 0;
                 }
                 // Destroyable:onHit (components.lua:275-291), health 1: destroyed, the beam passes
                 return (unsigned)(s - c.s_type0) < 4u ? 2 : 0;
               },
               [&](int, int, int, bool reached, int cell, bool touched) {
                 if (reached) {  // Zapper:onHit of an avatar standing there
                   const int pl = (int)(wd.sinfo[at(t.avatar_layer, cell)] >> 24) - 1;
                   if (pl >= 0) push_event(sc, MP_EVENT_ZAP, b + 1, pl + 1);
                 }
                 if (touched) mark[cell] |= 1;
               },
               b);
    // GraduatedSanctionsMarking:onHit for every avatar this beam reached
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

  // Zapper respawn: teleportToGroup(spawnGroup, aliveState).  A21: the new state's onAdd
  // (Avatar:onStateChange: the counters restart; the marking is told to come back) runs
  // before the cell's contact callbacks (the mushroom under the spawn point is eaten).
  const int mark_pos = my * W + mx;   // where this avatar's marking is, or waits
  const int rcell = resolve_respawns(t, wd, tail, a, want_respawn, order_resp, alive_state,
                                     (uint32_t)step, frame, ep, k0, k1);
  respawned = rcell >= 0;
  if (__ballot(respawned) != 0ull) {
    int eat_ty = -1;
    if (respawned) {
      freeze = 0; removal = 0;   // Avatar:onStateChange (avatar_library.lua:430-453)
      push_event(sc, MP_EVENT_SET_SANCTIONING_LEVEL, lane + 1, level);   // 'respawn': _setLevel(self._level)
      const int s = at(c.live_layer, rcell);
      if (s != 0) eat_ty = s - c.s_type0;
    }
    wsync();
    for (int r = 0; r < P; ++r) {
      const int p = rdlane(order_resp, r);
      const int T = rdlane(eat_ty, p);
      if (T >= 0) eat(p, T, rdlane(rcell, p), true, r);
    }
  }

  // end of flush 1: the resetToInitialLevel _setLevel, the perished mushrooms
  if (is_av && mark_reset && mstate > 0) {
    mstate = 1; at(c.mark_layer, mark_pos) = (uint8_t)c.s_mark[0];
  }
  // ... of a marking that never came back: onto the map at its transform — where its avatar
  // respawned — if nothing lies there (objects in creation order: two may want one cell)
  if (const unsigned long long stray = __ballot(is_av && mark_reset && mstate == 0)) {
    for (int p = 0; p < P; ++p) {
      if (lane == p && ((stray >> p) & 1ull) && at(c.mark_layer, mark_pos) == 0) {
        mstate = 1; at(c.mark_layer, mark_pos) = (uint8_t)c.s_mark[0];
      }
      wsync();
    }
  }
#pragma unroll
  for (int k = 0; k < kShroomPerLane; ++k)
    if ((perish_bits >> k) & 1u) at(c.live_layer, sites.site[k]) = 0;
  wsync();

  // ---- flush 2.  Markings: 'die' queued the wait state at the start of flush 1, a zap's
  // _setLevel later (the level lands last: an orphan), a respawn's _setLevel + teleport last.
  if (died && mstate > 0) { at(c.mark_layer, mark_pos) = 0; mstate = 0; }
  if (is_av && mark_level_pending > 0) {
    mstate = mark_level_pending;
    at(c.mark_layer, mark_pos) = (uint8_t)(c.s_mark[0] + (mark_level_pending - 1) * (c.s_mark[1] - c.s_mark[0]));
  }
  wsync();
  if (__ballot(respawned) != 0ull)
    for (int r = 0; r < P; ++r) {
      const int p = rdlane(order_resp, r);
      if (rdlane((int)respawned, p) == 0) continue;
      const int D = rdlane(mark_pos, p), A = rdlane(rcell, p);
      const int ms = rdlane(mstate, p), lv = rdlane(level, p);
      const int s_lv = c.s_mark[0] + (lv - 1) * (c.s_mark[1] - c.s_mark[0]);
      const int at_d = at(c.mark_layer, D), at_a = at(c.mark_layer, A);
      int now = 0;   // its state after the flush
      if (ms > 0 || at_d == 0) {
        // it is (an orphan) or comes back (nothing else there) where it waited, at the
        // avatar's level; then it teleports to its avatar — or stays behind, connected at a
        // distance, when another marking lies on the spawn cell
        now = lv;
        const bool follows = A == D || at_a == 0;
        if (lane == p) {
          if (follows) {
            at(c.mark_layer, D) = 0; at(c.mark_layer, A) = (uint8_t)s_lv;
            mx = A % W; my = A / W;
          } else {
            at(c.mark_layer, D) = (uint8_t)s_lv;
          }
        }
      } else if (lane == p) {
        mx = A % W; my = A / W;   // (the teleport of a piece that is off the grid: its transform)
      }
      if (lane == p) mstate = now;
      wsync();
    }

  // mushrooms: the setStates queued by the callbacks of flush 1; one frame older
#pragma unroll
  for (int k = 0; k < kShroomPerLane; ++k) {
    const int cell = sites.site[k];
    if (cell < 0) continue;
    const int m = mark[cell];
    if (m & 1) at(c.live_layer, cell) = 0;
    else if (m >> 1) {
      at(c.live_layer, cell) = (uint8_t)(c.s_type0 + (m >> 1) - 1);
      at(c.plane_age, cell) = 0;
    }
    mark[cell] = 0;
    if (at(c.live_layer, cell) != 0) {
      const int age = at(c.plane_age, cell);
      if (age < 255) at(c.plane_age, cell) = (uint8_t)(age + 1);
    }
  }
  wsync();

  const unsigned long long badb = __ballot(bad != 0);
  const int done = is_reset ? 0 : !(cont && step < t.max_frames);
  if (lane == 0) {
    tail->step = step;
    tail->frame = frame + 1;
    tail->cont = cont;
    tail->done = done;
    tail->aux_count = n_wait;   // this frame's potential sites (the Lua's set)
    if (!is_reset) { tail->ctr[0]++; tail->ctr[1] += (uint32_t)P; tail->ctr[7] += __popcll(badb); }
  }
  // (a statistic: markings that end the frame on the map away from their living avatar)
  if (const unsigned long long away = __ballot(is_av && a.alive && mstate > 0 && my * W + mx != a.y * W + a.x))
    if (lane == 0 && !is_reset) tail->ctr[5] += (uint32_t)__popcll(away);
  a.ctimer = mx;
  if (lane < MP_MAX_PLAYERS) {
    tail->flag0[lane] = (uint8_t)mstate; tail->flag1[lane] = (uint8_t)my;
    tail->freeze[lane] = (uint8_t)freeze; tail->removal[lane] = (uint8_t)removal;
    tail->aflags[lane] = (uint8_t)(mov_allowed | (disallow << 1));
    tail->nozap[lane] = (uint8_t)nozap; tail->level[lane] = (uint8_t)level;
    tail->tsince[lane] = (uint8_t)tsince;
  }
  const int step_type = is_reset ? 0 : (done ? 2 : 1);
  finish(t, wd, tail, a, 0.0, c.zap.cooldown, step_type, out, kOrders);
}

}  // namespace stepk

#endif  // MP_STEP_MUSHROOM_H_
