// step_matrix.h — one environment step (or episode start) of one *_in_the_matrix
// world by one wavefront (shape: step_clean_up.h / step_common.h).
//
// Substrate rules restated here (reference: configs/substrates/
// <game>_in_the_matrix__<variant>.py, the_matrix.py; lua/levels/the_matrix/
// components.lua; lua/modules/avatar_library.lua):
//   Resource           components.lua:32-129    pick-up on contact, regeneration
//   Destroyable        :132-185                 three zaps destroy a resource
//   TheMatrix          :188-290                 inventories, readiness, indicators
//   SpawnResourcesWhenAllPlayersZapped :293-321
//   GameInteractionZapper :328-937              beam, interaction, payoff, freeze,
//                                               effects scheduled behind the freeze
//   Taste / InteractionTaste / DyadicRole :966-1058
//   ReadyToInteractMarker :1066-1097 + AvatarConnector avatar_library.lua:884-945
//
// Record layout behind the L render planes:
//   plane A  per cell: health (bits 0-1) | 1 + class of the resource of this
//            episode (bits 2-3, 0 = none) | visible (bit 4)
//   plane B  per cell: frames since the resource's last state change, saturating
//   MxPlayer[16]  per player: the two rewards of the interaction whose effects
//            this player's component holds, and the inventory
// Tail bytes in use (besides position / orientation / alive / change frame):
//   ztimer  _coolingTimer         freeze  Avatar._freezeCounter
//   aflags  bit 0 _movementAllowed
//   flag0   marker: 0 = off the grid, else 1 + the indicator its state shows
//   ctimer / nozap   marker x / y (kept while it is off the grid)
//   flag1   indicator (bits 0-2) | collectedAtLeastOne (3) | effects pending (4)
//           | row player won (5) | _endEpisodeOnNextFrame (6)
//   level   _framesTillScheduledEffects + 1      tsince  colour interval
//   removal row | col << 4 of the pending effects
//
// The readiness marker is a piece of its own (it can end up detached from its
// avatar: a blocked setState on respawn, A14 follow-moves afterwards), so its
// position lives in registers next to the avatar's and every occupancy test on
// the overlay layer is a ballot over those registers.  Beams are evaluated one
// at a time in visiting order: an interaction changes what the next beam meets.
#ifndef MP_STEP_MATRIX_H_
#define MP_STEP_MATRIX_H_

#include "step_common.h"

namespace stepk {

constexpr int kMxSitesPerLane = 2;   // mp_create admits at most 128 site entries
constexpr int kMxMaxR = 3;

// cell | class << 16 of site k * 64 + lane, or -1
struct MatrixSites { int site[kMxSitesPerLane]; };

__device__ inline MatrixSites load_sites(const MatrixTables& c, int lane) {
  MatrixSites s;
#pragma unroll
  for (int k = 0; k < kMxSitesPerLane; ++k) {
    const int i = k * 64 + lane;
    s.site[k] = i < c.n_site ? (c.site_cells[i] | (c.site_class[i] << 16)) : -1;
  }
  return s;
}

struct MxPlayer {
  double row_reward, col_reward;
  uint16_t inv[4];
  uint32_t pad[2];
};
static_assert(sizeof(MxPlayer) == 32, "MxPlayer layout");

// TheMatrix:getColorInterval (components.lua:282-290).  The reference asserts
// that an interval matches; mp_create refuses a pack whose intervals leave a gap
// inside the range its matrix can pay; a reward on the range's very end (the
// stock intervals are half-open there) is reported through fault word 8.
__device__ inline int color_interval(const DevTables& t, const MatrixTables& c, double reward,
                                     int w) {
  int idx = -1;
  for (int k = c.n_intervals - 1; k >= 0; --k)
    if (c.interval[2 * k] <= reward && reward < c.interval[2 * k + 1]) idx = k;
  if (idx < 0) {   // the reference's assert: reported by the next synchronising call
    atomicCAS(&t.fault[8], 0u, (uint32_t)w + 1u);
    idx = 0;
  }
  return idx;
}

__device__ inline void step_world(const DevTables& t, const MatrixTables& c,
                                  const MatrixSites& sites, const World& wd, const Action& act,
                                  const StepArgs& args) {
#pragma clang fp contract(off)   // the payoffs are compared bit for bit with plain C
  const int lane = wd.lane, w = wd.w;
  const StepOutputs& out = args.out;
  Scratch* sc = wd.sc;
  uint8_t* grid = wd.rec;
  uint8_t* mark = wd.mark;   // per cell: bit 0 collected, bit 1 destroyed (this frame)
  WorldTail* tail = reinterpret_cast<WorldTail*>(wd.rec + t.grid_pad);
  MxPlayer* players = reinterpret_cast<MxPlayer*>(wd.rec + c.player_block);
  const int P = t.P, HW = t.H * t.W, W = t.W, R = c.R;
  const bool is_av = lane < P;
  auto at = [&](int layer, int cell) -> uint8_t& { return grid[layer * HW + cell]; };
  auto site_cell = [&](int k) { return sites.site[k] < 0 ? -1 : (sites.site[k] & 0xffff); };
  auto site_cls = [&](int k) { return sites.site[k] >> 16; };
  auto visible_state = [&](int cls1) { return (int)((c.s_visible_packed >> (8 * (cls1 - 1))) & 255u); };
  auto mark_state = [&](int m) { return (int)((c.s_mark_packed >> (8 * (m - 1))) & 255ull); };

  const OrderStreams kOrders = {RS_SHUFFLE_MOVE, RS_SHUFFLE_ZAP, RS_SHUFFLE_RESPAWN, 0, 3};   // the updater groups shuffled per frame (A1)

  const int what = dispatch(t, tail, lane, w, args.reset_mask, args.mode, args.auto_reset, out);
  if (what == 0) return;

  // per-player constants (the pack, global memory: one 16-byte and one 32-byte load)
  int taste_class = -1, role = -1;
  if (is_av) {
    const int4 pi = reinterpret_cast<const int4*>(c.player_i32)[lane];
    taste_class = pi.x; role = pi.w;
  }

  Av a;
  int freeze = 0, mstate = 0, mx = 0, my = 0, till = -1;
  // the small per-player variables share one register (flag1 / removal / tsince /
  // aflags of the tail; see the file comment)
  struct {
    uint32_t ind : 3, collected : 1, fx_pending : 1, fx_row_won : 1, end_next : 1,
        mov_allowed : 1, fx_row : 4, fx_col : 4, color : 3;
  } fl = {0, 0, 0, 0, 0, 1, 0, 0, 0};
  int inv[kMxMaxR] = {0, 0, 0};
  // latest_interaction_inventories go straight to the observation: -1 (0 at the
  // episode start) now, the two inventories when an interaction is resolved
  int interacted = 0;
  // binary cumulants of the step (components.lua:808-853; MP_OBS_MATRIX_CUMULANTS):
  // bit 0 interacted, 1 + 3k collected, 2 + 3k destroyed (from sc->flags), 3 + 3k argmax
  uint32_t cum = 0;
  auto report = [&](int s2, int k, double v) {
    out.interaction[(((size_t)w * P + lane) * 2 + s2) * R + k] = v;
  };
  const int inv0 = c.zero_inventory ? 0 : 1;
  const int alive_state = is_av ? t.alive_state[lane] : 0;
  int step_type, live_sites = 0;
  uint32_t k0, k1, ep;

  // beam sprites of the previous frame disappear (grid:update start)
  clear_bytes(grid, c.beam_layer * HW, HW, lane);

  if (what == 1) {
    // ---- api:start (api_factory.lua:85-102)
    k0 = (uint32_t)tail->seed; k1 = (uint32_t)(tail->seed >> 32);
    ep = tail->episode;
    wsync();
    const int gvec = (t.L * HW + 15) >> 4;
    for (int i = lane; i < gvec; i += 64)
      reinterpret_cast<uint4*>(grid)[i] = reinterpret_cast<const uint4*>(t.init_grid)[i];
    wsync();
    for (int i = t.L * HW + lane; i < t.grid_pad; i += 64) grid[i] = 0;   // hidden planes, players
    if (lane == 0) {
      tail->episode = ep + 1;
      tail->step = 0; tail->frame = 1; tail->done = 0; tail->cont = 1;
      tail->started = 1; tail->group_change = 0;
      tail->ctr[2]++;
    }
    wsync();
    apply_map_choices(t, grid, lane, ep, k0, k1);
    // Destroyable:reset; a resource of a 'choice' cell that is not in this
    // episode's map has no entry in plane A ("absent": every rule skips it)
#pragma unroll
    for (int k = 0; k < kMxSitesPerLane; ++k) {
      const int cell = site_cell(k);
      if (cell < 0) continue;
      if (at(c.res_layer, cell) == visible_state(site_cls(k))) {
        at(c.plane_a, cell) = (uint8_t)(c.initial_health | (site_cls(k) << 2) | 16);
        at(c.plane_b, cell) = 1;   // (after the grid:update of api:start)
      }
    }
    wsync();
    spawn_avatars(t, grid, lane, ep, k0, k1, a);
    if (is_av) {
      // TheMatrix:reset, GameInteractionZapper:start; AvatarConnector:postStart
      // puts the marker on its avatar, state notReady
#pragma unroll
      for (int k = 0; k < kMxMaxR; ++k) inv[k] = k < R ? inv0 : 0;
      mstate = 1; mx = a.x; my = a.y;
      at(c.mark_layer, my * W + mx) = (uint8_t)mark_state(1);
      push_event(sc, MP_EVENT_AVATAR_STARTED, 0, 0);
      // a fresh tensor: zeros until the first GameInteractionZapper:update
      for (int k = 0; k < R; ++k) { report(0, k, 0.0); report(1, k, 0.0); }
    }
    live_sites = 0;
#pragma unroll
    for (int k = 0; k < kMxSitesPerLane; ++k)
      live_sites += __popcll(__ballot(site_cell(k) >= 0 &&
          ((at(c.plane_a, site_cell(k) >= 0 ? site_cell(k) : 0) >> 2) & 3) == site_cls(k)));
    step_type = 0;
  } else {
    // ================= api:advance =================
    k0 = (uint32_t)tail->seed; k1 = (uint32_t)(tail->seed >> 32);
    ep = tail->episode - 1;
    const int step = tail->step + 1, frame = tail->frame;
    load_avatars(tail, lane, a);
    if (lane < MP_MAX_PLAYERS) {
      freeze = tail->freeze[lane]; fl.mov_allowed = tail->aflags[lane] & 1;
      mstate = tail->flag0[lane]; mx = tail->ctimer[lane]; my = tail->nozap[lane];
      const int f1 = tail->flag1[lane];
      fl.ind = f1 & 7; fl.collected = (f1 >> 3) & 1; fl.fx_pending = (f1 >> 4) & 1;
      fl.fx_row_won = (f1 >> 5) & 1; fl.end_next = (f1 >> 6) & 1;
      till = (int)tail->level[lane] - 1; fl.color = tail->tsince[lane];
      fl.fx_row = tail->removal[lane] & 15; fl.fx_col = tail->removal[lane] >> 4;
#pragma unroll
      for (int k = 0; k < kMxMaxR; ++k) inv[k] = players[lane].inv[k];
    }
    a.ctimer = 0;
    if (is_av)
      for (int k = 0; k < R; ++k) { report(0, k, -1.0); report(1, k, -1.0); }
    wsync();
    auto draw = [&](int stream, uint32_t index) {
      return philox4x32_10(index, (uint32_t)stream, (uint32_t)step, ep, k0, k1);
    };
    // ---- BaseSimulation:update: Avatar:update (avatar_library.lua:334-355);
    // GameInteractionZapper:update resets the interaction report (inter = -1)
    if (is_av) {
      if (freeze == 1) fl.mov_allowed = 1;
      if (freeze > 0) freeze--;
    }
    // ---- updaters, priority descending; they read the pre-flush state
    int cont = tail->cont;
    // 900 endEpisodeIfApplicable
    if (__ballot(is_av && fl.end_next != 0) != 0) cont = 0;
    // 890 resetSimultaneousInteractionBlocker
    int iflag = 0;
    int orders[4];
    step_orders(tail, lane, P, kOrders, (uint32_t)step, ep, k0, k1, orders);
    const int order_move = orders[0], order_zap = orders[1], order_resp = orders[2];
    // 150 Avatar move: nothing while movement is disallowed
    const int a_move = fl.mov_allowed ? act.move : 0, a_turn = fl.mov_allowed ? act.turn : 0;
    // 140 zap (components.lua:400-424): gated by movement too
    bool fire = false;
    if (is_av && fl.mov_allowed && a.alive && c.cooldown >= 0) {
      if (a.ztimer > 0) a.ztimer--;
      else if (act.fire0 == 1) { a.ztimer = c.cooldown; fire = true; }
    }
    // 135 respawn: state = waitState, startFrame = framesTillRespawn
    const bool want_respawn = is_av && !a.alive && frame - a.achange >= c.respawn_frames;
    // 100 StochasticIntervalEpisodeEnding: _t == step + 1
    if (c.has_ee && frame >= c.ee_min_frames && (step + 1) % c.ee_interval == 0)
      if (philox_u53(draw(RS_EPISODE_END, 0)) < c.thr_ee) cont = 0;
    // 100 Resource maybeRespawn (components.lua:84-101), 7 SpawnResourcesWhenAll
    // PlayersZapped (:303-321): decided on the pre-flush state, applied behind the
    // moves, beams and respawns of this flush
    const unsigned long long alive_pre = __ballot(is_av && a.alive);
    uint32_t appear = 0;
#pragma unroll
    for (int k = 0; k < kMxSitesPerLane; ++k) {
      const int cell = site_cell(k), i = k * 64 + lane;
      if (cell < 0) continue;
      const int A = at(c.plane_a, cell);
      if (((A >> 2) & 3) != site_cls(k) || ((A >> 4) & 1)) continue;   // absent, or visible
      bool back = c.spawn_all && alive_pre == 0;
      if (!back && at(c.plane_b, cell) >= c.regen_delay &&
          philox_u53(draw(RS_REGROW, (uint32_t)i)) < c.thr_regen &&
          at(t.avatar_layer, cell) == 0)
        back = true;
      if (back) appear |= 1u << k;
    }
    // 4 applyScheduledEffects, state = aliveState (components.lua:439-450)
    bool die = false;
    const bool apply = is_av && a.alive && till == 0;
    const bool count_down = is_av && a.alive && till > 0;
    unsigned long long owners = __ballot(apply && fl.fx_pending != 0);
    while (owners != 0) {
      const int p = __ffsll((long long)owners) - 1;
      owners &= owners - 1;
      const int row = rdlane((int)fl.fx_row, p), col = rdlane((int)fl.fx_col, p), row_won = rdlane((int)fl.fx_row_won, p);
      const double rr = players[p].row_reward, cr = players[p].col_reward;   // (uniform p)
      // sendRewardsToBothInteractants (:527-549): the ZAPPED player's
      // InteractionTaste prices both rewards, on the inventories as they are now
      const int tasty = c.player_i32[4 * p + 1], zero_default = c.player_i32[4 * p + 2];
      const double extra = c.player_f64[4 * p + 2];
      auto taste = [&](double reward) {   // getExtraRewardForInteraction (:1019-1039)
        if (tasty > 0) {
          if (zero_default) reward = 0.0;
          const int amount = tasty == 1 ? inv[0] : tasty == 2 ? inv[1] : inv[2];
          bool maximal = true;
          for (int idx = 1; idx <= R; ++idx)
            if (idx != tasty) maximal = amount > (idx == 1 ? inv[0] : idx == 2 ? inv[1] : inv[2]);
          if (maximal) return reward + extra;
        }
        return reward;
      };
      if (rr > c.reward_floor && lane == row) a.reward += taste(rr);
      if (cr > c.reward_floor && lane == col) a.reward += taste(cr);
      const int winner = row_won ? row : col, loser = row_won ? col : row;
      if ((c.reset_loser && lane == loser) || (c.reset_winner && lane == winner)) {
#pragma unroll
        for (int k = 0; k < kMxMaxR; ++k) inv[k] = k < R ? inv0 : 0;
        fl.collected = 0;
      }
      if ((c.loser_dies && lane == loser) || (c.winner_dies && lane == winner)) die = true;
      if (lane == p) fl.fx_pending = 0;
    }
    if (apply) {
      fl.ind = 0; till = -1;
      if (c.end_on_first) fl.end_next = 1;
    } else if (count_down) {
      till--;
      fl.ind = 2 + fl.color;
    }
    // 2 ReadyToInteractMarker displayReadiness (:1081-1096)
    const int want_m = (is_av && a.alive) ? fl.ind + 1 : 0;

    // ---- flush 1, FIFO
    // moves (avatar_library.lua:155-203): the avatar and its connected marker move
    // as a unit (A14); every target test is made in visiting order
    if (is_av && a_turn != 0) a.ori = (a.ori + a_turn + 4) & 3;   // off-grid pieces turn too
    const bool wants = is_av && a.alive && a_move != 0;
    int tx = a.x, ty = a.y, tmx = mx, tmy = my;
    bool target_ok = false;
    if (wants) {
      const int dir = (a.ori + a_move - 1) & 3;
      const int dx = dir_dx(dir), dy = dir_dy(dir);
      if (step_cell(t, tx, ty, dx, dy)) {
        const int s = at(t.avatar_layer, ty * W + tx);
        target_ok = s == 0 || (wd.sinfo[s] >> 24) != 0;   // walls; avatars are decided in order
      }
      if (mstate > 0 && !step_cell(t, tmx, tmy, dx, dy)) target_ok = false;
    }
    const int old_cell = a.y * W + a.x, old_mcell = my * W + mx;
    bool moved = false;
    const int try_move = (int)(wants && target_ok);
    if (__ballot(try_move != 0) != 0) {
      for (int r = 0; r < P; ++r) {
        const int p = rdlane(order_move, r);
        if (rdlane(try_move, p) == 0) continue;
        const int ptx = rdlane(tx, p), pty = rdlane(ty, p);
        const int pmx = rdlane(tmx, p), pmy = rdlane(tmy, p), pm_on = rdlane(mstate, p);
        const bool occ_a = __ballot(is_av && a.alive && a.x == ptx && a.y == pty) != 0;
        const bool occ_m = __ballot(is_av && mstate > 0 && mx == pmx && my == pmy) != 0;
        if (lane == p && !occ_a && !(pm_on > 0 && occ_m)) {
          a.x = ptx; a.y = pty; moved = true;
          if (mstate > 0) { mx = pmx; my = pmy; }
        }
      }
    }
    if (moved) {
      at(t.avatar_layer, old_cell) = 0;
      if (mstate > 0) at(c.mark_layer, old_mcell) = 0;
    }
    wsync();
    if (moved) {
      at(t.avatar_layer, a.y * W + a.x) = (uint8_t)alive_state;
      if (mstate > 0) at(c.mark_layer, my * W + mx) = (uint8_t)mark_state(mstate);
    }
    wsync();
    // Resource:onEnter (components.lua:54-82) on the cell the avatar is in now
    // (A3b: a blocked move re-enters its own cell)
    if (wants) {
      const int cell = a.y * W + a.x;
      const int A = at(c.plane_a, cell);
      if ((A >> 4) & 1) {
        const int cls = (A >> 2) & 3;
        if (cls == 1) inv[0] = min(inv[0] + 1, 65535);
        else if (cls == 2) inv[1] = min(inv[1] + 1, 65535);
        else inv[2] = min(inv[2] + 1, 65535);
        fl.collected = 1;
        cum |= 2u << (3 * (cls - 1));   // setResourceCollectionCumulant (:120)
        if (fl.ind == 0) fl.ind = 1;
        mark[cell] |= 1;   // setState(waitState), next flush
        a.reward += c.player_f64[4 * lane + (cls == taste_class ? 0 : 1)];   // Taste (:985-990)
        push_event(sc, MP_EVENT_COLLECTED_RESOURCE, lane + 1, cls);
      }
    }
    wsync();

    // beams, one at a time in visiting order
    const BeamLane shape = beam_lane(c.shape, lane);
    const int firing = (int)(fire && a.alive);
    for (int r = 0; r < P; ++r) {
      const int b = rdlane(order_zap, r);
      if (rdlane(firing, b) == 0) continue;
      fire_beams(t, wd, tail, a, fire, shape, c.hit, false, c.beam_layer, c.s_beam, false,
                 [&](int s, int cell) {
                   const int pl = (int)(wd.sinfo[s] >> 24);
                   if (pl != 0) return 1 | (pl << 8);   // GameInteractionZapper:onHit always stops the beam
                   const uint32_t v = c.s_visible_packed;
                   if ((uint32_t)s != (v & 255u) && (uint32_t)s != ((v >> 8) & 255u) &&
                       (uint32_t)s != ((v >> 16) & 255u))
                     return 0;
                   // Destroyable:onHit (:154-172): stops the beam unless this hit destroys it
                   return ((at(c.plane_a, cell) & 3) - 1 != 0) ? 3 : 2;
                 },
                 [&](int, int, int, bool, int cell, bool touched) {
                   if (!touched) return;
                   int A = at(c.plane_a, cell);
                   int health = (A & 3) - 1;
                   if (health == 0) {
                     health = c.initial_health;
                     mark[cell] |= 2;
                     push_event(sc, MP_EVENT_DESTROYED_RESOURCE, b + 1, (A >> 2) & 3);
                     // setResourceDestructionCumulant of the zapper (:180)
                     atomicOr(&sc->flags[b >> 3], 1u << (4 * (b & 7) + ((A >> 2) & 3) - 1));
                   }
                   at(c.plane_a, cell) = (uint8_t)((A & ~3) | health);
                 },
                 b);
      // GameInteractionZapper:onHit (components.lua:720-759) of every avatar the
      // beam reached, in footprint order
      for (int j = 0; j < c.shape.n; ++j) {
        const int v = __builtin_amdgcn_readfirstlane((int)sc->victim[b][j]);
        if (v < 0) continue;
        // _preventExtraSimultaneousInteraction (:705-718)
        if (rdlane(iflag, v)) continue;
        if (lane == v) iflag = 1;
        if (rdlane(iflag, b)) continue;
        if (lane == b) iflag = 1;
        if (rdlane(till, v) >= 0) continue;   // frozen players cannot be zapped
        const int cv = rdlane((int)fl.collected, v), cb = rdlane((int)fl.collected, b);
        if (!cv && lane == b) a.reward += c.reward_unready;
        if (c.disallow_unready && !(cb && cv)) continue;
        if (lane == v || lane == b) cum |= 1u;   // _setInteractionCumulant (:768)
        int row = b, col = v;   // the zapper is the row player ...
        const int role_v = rdlane(role, v), role_b = rdlane(role, b);
        if (role_v >= 0 && role_b >= 0) {   // ... unless both carry a DyadicRole (:736-750)
          if (role_b == 1 && role_v == 0) { row = b; col = v; }
          else if (role_b == 0 && role_v == 1) { row = v; col = b; }
          else continue;
        }
        // ---- _resolve (components.lua:556-703); v's component holds the effects.
        // The inventories stay integers (they are small counts: every conversion
        // and sum below is exact) and the column profile is formed where it is
        // used: a third fewer live registers than two double[3] pairs.
        const int ri0 = rdlane(inv[0], row), ri1 = rdlane(inv[1], row), ri2 = rdlane(inv[2], row);
        const int ci0 = rdlane(inv[0], col), ci1 = rdlane(inv[1], col), ci2 = rdlane(inv[2], col);
        auto rinv = [&](int k) { return k == 0 ? ri0 : k == 1 ? ri1 : ri2; };
        auto cinv = [&](int k) { return k == 0 ? ci0 : k == 1 ? ci1 : ci2; };
        double rsum = 0.0, csum = 0.0;
#pragma unroll
        for (int k = 0; k < kMxMaxR; ++k)
          if (k < R) { rsum += (double)rinv(k); csum += (double)cinv(k); }
        double rp[kMxMaxR];
#pragma unroll
        for (int k = 0; k < kMxMaxR; ++k)
          rp[k] = rsum > 0.0 ? (double)rinv(k) / rsum : (double)rinv(k);
        // _computeInteractionRewards: (rowProfile . M) . colProfile, left to right
        double row_reward = 0.0, col_reward = 0.0;
#pragma unroll
        for (int jj = 0; jj < kMxMaxR; ++jj) {
          if (jj >= R) continue;
          double ta = 0.0, tb = 0.0;
#pragma unroll
          for (int i = 0; i < kMxMaxR; ++i) {
            if (i >= R) continue;
            ta += rp[i] * c.row_matrix[i * R + jj];
            tb += rp[i] * c.col_matrix[i * R + jj];
          }
          const double cpj = csum > 0.0 ? (double)cinv(jj) / csum : (double)cinv(jj);
          row_reward += ta * cpj;
          col_reward += tb * cpj;
        }
        row_reward = c.reward_multiplier * row_reward;
        col_reward = c.reward_multiplier * col_reward;
        // reportInteraction (:761-783): own inventory first
        if (lane == v || lane == b) {
          const bool self_is_row = lane == row;
          interacted = 1;
#pragma unroll
          for (int k = 0; k < kMxMaxR; ++k) {
            if (k >= R) continue;
            report(0, k, (double)(self_is_row ? rinv(k) : cinv(k)));
            report(1, k, (double)(self_is_row ? cinv(k) : rinv(k)));
          }
        }
        if (lane == v) push_event(sc, MP_EVENT_INTERACTION, row + 1, col + 1);
        // ... and the rest of the event's payload (:789-797): the inventories are
        // reportInteraction's above, the rewards go here, for both players
        if ((lane == row || lane == col) && out.interaction_rewards) {
          double* ir = out.interaction_rewards + ((size_t)w * P + lane) * 2;
          ir[0] = row_reward; ir[1] = col_reward;
        }
        if (lane == row || lane == col) {   // setArgMaxCumulants (:808-815): first maximal class
          const bool is_row = lane == row;
          const int i0 = is_row ? ri0 : ci0, i1 = is_row ? ri1 : ci1, i2 = is_row ? ri2 : ci2;
          int arg = 0, top = i0;
          if (R > 1 && i1 > top) { arg = 1; top = i1; }
          if (R > 2 && i2 > top) { arg = 2; top = i2; }
          if (top > 0) cum |= 8u << (3 * arg);
        }
        int row_won;
        if (row_reward > col_reward) row_won = 1;
        else if (row_reward == col_reward) {
          row_won = 1;
          if (c.random_tie)   // uniformReal(0, 1) <= 0.5
            row_won = philox_u53(draw(RS_TIE_BREAK, (uint32_t)v)) <= (1ull << 52);
        } else row_won = 0;
        if (lane == row || lane == col) till = c.freeze;
        if (lane == v) {
          fl.fx_pending = 1; fl.fx_row = row; fl.fx_col = col; fl.fx_row_won = row_won;
          players[lane].row_reward = row_reward; players[lane].col_reward = col_reward;
        }
        // as written (:648-651): a winning row player's inventory is also reset at once
        if (row_won && c.reset_winner && lane == row) {
#pragma unroll
          for (int k = 0; k < kMxMaxR; ++k) inv[k] = k < R ? inv0 : 0;
          fl.collected = 0;
        }
        if (c.freeze + 2 > 0 && (lane == row || lane == col)) {   // disallowMovementUntil
          fl.mov_allowed = 0; freeze = c.freeze + 2;
        }
        if (lane == row) fl.color = (uint32_t)color_interval(t, c, row_reward, w);
        if (lane == col) fl.color = (uint32_t)color_interval(t, c, col_reward, w);
      }
      wsync();
    }

    // respawns (135), PICK_RANDOM orientation; 'respawn' resets the freeze counter
    const int rcell = resolve_respawns(t, wd, tail, a, want_respawn, order_resp, alive_state,
                                       (uint32_t)step, frame, ep, k0, k1);
    const bool respawned = rcell >= 0;
    if (respawned) freeze = 0;
    // regenerated resources appear
#pragma unroll
    for (int k = 0; k < kMxSitesPerLane; ++k) {
      if (!((appear >> k) & 1u)) continue;
      const int cell = site_cell(k);
      at(c.plane_a, cell) |= 16;
      at(c.plane_b, cell) = 0;
      at(c.res_layer, cell) = (uint8_t)visible_state(site_cls(k));
    }
    // _avatarDies: the scheduled removals
    bool died = false;
    if (die && a.alive) {
      at(t.avatar_layer, a.y * W + a.x) = 0;
      a.alive = 0; a.achange = frame; died = true;
    }
    wsync();
    // the markers' setState of updater 2.  A marker on the grid changes state in
    // place; one that is off the grid (its avatar alive: an earlier placement was
    // blocked) tries its cell again, one whose avatar waits leaves the grid
    if (is_av && want_m > 0 && mstate > 0) {
      mstate = want_m;
      at(c.mark_layer, my * W + mx) = (uint8_t)mark_state(mstate);
    }
    {
      const bool rare = is_av && ((want_m > 0 && mstate == 0) || (want_m == 0 && mstate > 0));
      const unsigned long long rb = __ballot(rare);
      if (rb != 0) {
        for (int p = 0; p < P; ++p) {
          if (!((rb >> p) & 1ull)) continue;
          const int pmx = rdlane(mx, p), pmy = rdlane(my, p);
          const bool occ = __ballot(is_av && mstate > 0 && mx == pmx && my == pmy) != 0;
          if (lane == p) {
            if (want_m == 0) { at(c.mark_layer, my * W + mx) = 0; mstate = 0; }
            else if (!occ) { mstate = want_m; at(c.mark_layer, my * W + mx) = (uint8_t)mark_state(mstate); }
          }
        }
      }
    }
    wsync();

    // ---- flush 2: collected / destroyed resources wait
#pragma unroll
    for (int k = 0; k < kMxSitesPerLane; ++k) {
      const int cell = site_cell(k);
      if (cell < 0 || mark[cell] == 0) continue;
      if (((at(c.plane_a, cell) >> 2) & 3) != site_cls(k)) continue;
      at(c.plane_a, cell) &= ~16;
      at(c.plane_b, cell) = 0;
      at(c.res_layer, cell) = 0;
    }
    wsync();
#pragma unroll
    for (int k = 0; k < kMxSitesPerLane; ++k)
      if (site_cell(k) >= 0) mark[site_cell(k)] = 0;
    // AvatarConnector:avatarStateChange('respawn') (avatar_library.lua:923-934),
    // in the order the respawns were processed: setState(notReady) where the
    // marker was left, then teleport onto the avatar
    {
      const unsigned long long rsp = __ballot(respawned);
      if (rsp != 0) {
        for (int r = 0; r < P; ++r) {
          const int p = rdlane(order_resp, r);
          if (!((rsp >> p) & 1ull)) continue;
          const int pmx = rdlane(mx, p), pmy = rdlane(my, p);
          const bool occ = __ballot(is_av && lane != p && mstate > 0 && mx == pmx && my == pmy) != 0;
          if (lane == p) {
            if (mstate > 0) { mstate = 1; at(c.mark_layer, my * W + mx) = (uint8_t)mark_state(1); }
            else if (!occ) { mstate = 1; at(c.mark_layer, my * W + mx) = (uint8_t)mark_state(1); }
          }
          const int pax = rdlane(a.x, p), pay = rdlane(a.y, p);
          const bool occ2 = __ballot(is_av && lane != p && mstate > 0 && mx == pax && my == pay) != 0;
          if (lane == p) {
            if (mstate == 0) { mx = pax; my = pay; }   // a piece off the grid just takes the position
            else if (!occ2) {
              at(c.mark_layer, my * W + mx) = 0;
              mx = pax; my = pay;
              at(c.mark_layer, my * W + mx) = (uint8_t)mark_state(mstate);
            }
          }
        }
      }
    }
    // ... ('die'): the markers of the removed avatars leave the grid
    if (died && mstate > 0) { at(c.mark_layer, my * W + mx) = 0; mstate = 0; }
    wsync();

    // end of the frame: site ages
    live_sites = 0;
#pragma unroll
    for (int k = 0; k < kMxSitesPerLane; ++k) {
      const int cell = site_cell(k);
      bool present = false;
      if (cell >= 0) {
        const int A = at(c.plane_a, cell);
        present = ((A >> 2) & 3) == site_cls(k);
        if (present) {
          const int B = at(c.plane_b, cell);
          if (B < 255) at(c.plane_b, cell) = (uint8_t)(B + 1);
        }
        present = present && ((A >> 4) & 1);
      }
      live_sites += __popcll(__ballot(present));
    }
    const unsigned long long badb = __ballot(act.bad != 0);
    const unsigned long long hits = __ballot(is_av && interacted != 0);
    const int done = !(cont && step < t.max_frames);
    if (lane == 0) {
      tail->step = step;
      tail->frame = frame + 1;
      tail->cont = cont;
      tail->done = done;
      tail->ctr[0]++; tail->ctr[1] += (uint32_t)P; tail->ctr[7] += __popcll(badb);
      tail->ctr[5] += __popcll(hits);   // players that took part in an interaction
    }
    step_type = done ? 2 : 1;
  }
  // ---- registers -> record, observations
  if (lane == 0) tail->aux_count = live_sites;
  if (lane < MP_MAX_PLAYERS) {
    tail->freeze[lane] = (uint8_t)freeze;
    tail->aflags[lane] = (uint8_t)(fl.mov_allowed & 1);
    tail->flag0[lane] = (uint8_t)mstate;
    tail->flag1[lane] = (uint8_t)(fl.ind | (fl.collected << 3) | (fl.fx_pending << 4) | (fl.fx_row_won << 5) |
                                  (fl.end_next << 6));
    tail->level[lane] = (uint8_t)(till + 1);
    tail->tsince[lane] = (uint8_t)fl.color;
    tail->removal[lane] = (uint8_t)(fl.fx_row | (fl.fx_col << 4));
    tail->nozap[lane] = (uint8_t)my;
    players[lane].inv[0] = (uint16_t)inv[0]; players[lane].inv[1] = (uint16_t)inv[1];
    players[lane].inv[2] = (uint16_t)inv[2];
  }
  a.ctimer = mx;   // (finish stores ctimer)
  if (is_av) {
    const size_t o = (size_t)w * P + lane;
    for (int k = 0; k < R; ++k)
      out.inventory[o * R + k] = (double)(k == 0 ? inv[0] : k == 1 ? inv[1] : inv[2]);
    if (out.cumulants) {
      const uint32_t destroyed = (sc->flags[lane >> 3] >> (4 * (lane & 7))) & 7u;
      for (int k = 0; k < R; ++k)
        if ((destroyed >> k) & 1u) cum |= 4u << (3 * k);
      const int C = 1 + 3 * R;
      for (int k = 0; k < C; ++k) out.cumulants[o * C + k] = (double)((cum >> k) & 1u);
    }
  }
  const int ztimer = a.ztimer;
  finish(t, wd, tail, a, 0.0, c.cooldown > 0 ? c.cooldown : 1, step_type, out, kOrders);
  // GameInteractionZapper:readyToShoot (:914-917) does not look at the avatar's state
  if (is_av) out.ready[(size_t)w * P + lane] = 1.0 - (double)ztimer / (double)c.cooldown;
}

}  // namespace stepk

#endif  // MP_STEP_MATRIX_H_
