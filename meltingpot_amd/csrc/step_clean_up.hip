// step_clean_up.hip — one environment step (or episode start) of N clean_up
// worlds, one wavefront per world.
//
// Replaces, for the clean_up substrate, the reference's per-step path
//   api:advance            lua/modules/api_factory.lua:104-111
//   BaseSimulation:update  lua/modules/base_simulation.lua:476-486
//   grid:update(random)    dmlab2d (documented cycle: docs/advanced.md:33-52)
// and the Lua component callbacks it drives (cited at each block below), and
// the episode start path api:start (api_factory.lua:85-102,
// base_simulation.lua:396-471).
//
// Execution shape (v2).  A 64-lane workgroup streams its world record (grid
// planes + tail, ≈6 KB) from HBM into LDS with 16-byte lane loads, steps it
// there and streams it back.  Nothing in the step is resolved by a serial lane:
//   * lane p (< P) owns avatar p — position, orientation, timers, reward live in
//     its registers for the whole step;
//   * site work (122 AppleGrow draws, 147 dirt sites, 167 water pieces, plane
//     clears) is spread over the 64 lanes, with ballots for set selection;
//   * the frame's shuffled visiting orders (A1) are drawn one Philox call per
//     lane and applied with lane exchanges;
//   * moves and respawns are resolved in visiting order with one ballot per
//     avatar ("is any live avatar standing on my target?") instead of grid
//     reads, so the ordered phase touches no memory at all;
//   * beams: lane (b, j) evaluates footprint cell j of avatar b's beam; a
//     ballot of the "stops the beam" predicate against a per-cell predecessor
//     mask gives every cell's reached/not-reached in one step (63 lanes for 7
//     avatars x 9 cells).  Beams never change state inside the flush (their
//     effects are queued to the next flush), so all beams evaluate at once.
// v1 ran the ordered phase on lane 0 against LDS and took ~67 us for 4096
// worlds (profiles/r01_v1_baseline.md): pure LDS round-trip latency.
#include "mp_common.h"

namespace {

constexpr int kDx[4] = {0, 1, 0, -1};  // N E S W; N = decreasing y
constexpr int kDy[4] = {-1, 0, 1, 0};  // (component_library.lua:379-386)

enum { HIT_ZAP = 0, HIT_CLEAN = 1 };

// Per-wave scratch placed after the world record in LDS.
struct Scratch {
  uint8_t hit_block[256];
  int8_t splayer[256];
  int8_t victim[MP_MAX_PLAYERS][16];  // avatar hit by cell j of avatar b's zap beam
  uint32_t zapped_mask;
  int32_t pad;
  // followed by uint8_t mark[H*W]: dirt cells hit by a clean beam this frame
};

__device__ inline bool step_cell(const DevTables& t, int& x, int& y, int dx, int dy) {
  x += dx; y += dy;
  if (t.topology == 1) {  // TORUS
    x = ((x % t.W) + t.W) % t.W; y = ((y % t.H) + t.H) % t.H;
    return true;
  }
  return x >= 0 && x < t.W && y >= 0 && y < t.H;
}

// Lane-parallel count of the sites with pred true (ascending site order kept in
// the ballot masks); wave-uniform result.
template <class Pred>
__device__ int count_sites(int lane, int n, Pred pred, unsigned long long* masks) {
  int total = 0;
  const int chunks = (n + 63) >> 6;
  for (int ch = 0; ch < chunks; ++ch) {
    const int site = ch * 64 + lane;
    const bool v = site < n && pred(site);
    const unsigned long long m = __ballot(v);
    masks[ch] = m;
    total += __popcll(m);
  }
  return total;
}
__device__ int kth_site(const unsigned long long* masks, int chunks, int k) {
  for (int ch = 0; ch < chunks; ++ch) {
    unsigned long long m = masks[ch];
    const int pc = __popcll(m);
    if (k < pc) {
      for (int i = 0; i < k; ++i) m &= m - 1;
      return ch * 64 + __ffsll((long long)m) - 1;
    }
    k -= pc;
  }
  return -1;
}

// A1: the engine visits the pieces of an updater group in a freshly shuffled
// order every frame; forward Fisher-Yates, one draw per position.  Lane i draws
// position i's partner; the swaps are applied with lane exchanges.  Returns, in
// lane k, the avatar visited k-th.
__device__ int shuffled_order(int lane, int P, int stream, uint32_t step, uint32_t k0,
                              uint32_t k1) {
  int j = lane;
  if (lane + 1 < P)
    j = lane + (int)philox_bounded(
        philox4x32_10((uint32_t)lane, (uint32_t)stream, step, 0u, k0, k1),
        (uint32_t)(P - lane));
  int item = lane;
  for (int i = 0; i + 1 < P; ++i) {
    const int ji = __shfl(j, i);
    const int vi = __shfl(item, i), vj = __shfl(item, ji);
    if (lane == i) item = vj;
    else if (lane == ji) item = vi;
  }
  return item;
}

__global__ __launch_bounds__(64) void k_step_clean_up(
    DevTables t, CleanUpTables c, uint8_t* __restrict__ state,
    const int32_t* __restrict__ actions, const uint8_t* __restrict__ reset_mask,
    int mode, int auto_reset, StepOutputs out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  uint8_t* gw = state + (size_t)w * t.world_stride;
  const int nvec = t.world_stride >> 4;
  for (int i = lane; i < nvec; i += 64)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(gw)[i];
  Scratch* sc = reinterpret_cast<Scratch*>(smem + t.world_stride);
  for (int s = lane; s < 256; s += 64) {
    sc->hit_block[s] = s < t.nstates ? (uint8_t)t.state_hit_block[s] : 0;
    sc->splayer[s] = s < t.nstates ? t.state_player[s] : (int8_t)-1;
  }
  uint8_t* mark = reinterpret_cast<uint8_t*>(sc + 1);
  for (int i = lane; i < t.H * t.W; i += 64) mark[i] = 0;
  __syncthreads();
  uint8_t* grid = smem;
  WorldTail* tail = reinterpret_cast<WorldTail*>(smem + t.grid_pad);
  const int P = t.P, HW = t.H * t.W, W = t.W;
  const bool is_av = lane < P;
  auto at = [&](int layer, int cell) -> uint8_t& { return grid[layer * HW + cell]; };

  bool do_reset;
  if (mode == STEP_MODE_RESET) {
    do_reset = reset_mask ? reset_mask[w] != 0 : true;
    if (!do_reset) return;
  } else {
    if (!tail->started) return;  // never reset: nothing to step
    do_reset = tail->done && auto_reset;
    if (tail->done && !auto_reset) {  // frozen after LAST until mp_reset
      if (is_av) out.reward[w * P + lane] = 0.0;
      if (lane == 0) { out.collective[w] = 0.0; out.step_type[w] = 2; out.discount[w] = 0.0; }
      return;
    }
  }

  // avatar registers (lane p < P)
  int ax = 0, ay = 0, aori = 0, alive = 0, ztimer = 0, ctimer = 0, achange = 0;
  double reward = 0.0, aux0 = 0.0;
  int step_type;

  if (do_reset) {
    // ---- api:start(episode, seed) (api_factory.lua:85-102); every reset of a
    // world uses seed + #earlier resets (builder.py:177-181).
    const uint64_t seed = tail->seed + tail->episode;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    __syncthreads();
    const int gvec = t.grid_pad >> 4;
    for (int i = lane; i < gvec; i += 64)
      reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(t.init_grid)[i];
    if (lane == 0) {
      tail->episode++;
      tail->step = 0; tail->frame = 1; tail->done = 0; tail->cont = 1;
      tail->started = 1;
      tail->aux_count = c.n_dirt_init;  // DirtTracker:postStart (:103-116)
      tail->group_change = 0;
      tail->ctr[2]++;
    }
    // _avatarStart: groupShuffledWithCount(random, spawnGroup, numAvatars)
    // (base_simulation.lua:416-421): partial Fisher-Yates over the group's
    // pieces in creation order; avatar i takes the i-th sampled point.
    // Lane k holds spawn point k (n_spawn <= 64 here; checked at create).
    int item = lane < t.n_spawn ? t.spawn_cells[lane] : 0;
    int j = lane;
    if (is_av)
      j = lane + (int)philox_bounded(
          philox4x32_10((uint32_t)lane, RS_START_SPAWN, 0u, 0u, k0, k1),
          (uint32_t)(t.n_spawn - lane));
    for (int i = 0; i < P; ++i) {
      const int ji = __shfl(j, i);
      const int vi = __shfl(item, i), vj = __shfl(item, ji);
      if (lane == i) item = vj;
      else if (lane == ji) item = vi;
    }
    __syncthreads();
    if (is_av) {
      // Avatar:start (avatar_library.lua:288-320): random:choice(_COMPASS)
      aori = (int)philox_bounded(
          philox4x32_10((uint32_t)lane, RS_START_ORIENT, 0u, 0u, k0, k1), 4u);
      ax = item % W; ay = item / W; alive = 1;
      at(t.avatar_layer, item) = (uint8_t)t.alive_state[lane];
    }
    if (lane < MP_MAX_PLAYERS) { tail->flag0[lane] = 0; tail->flag1[lane] = 0; }
    // Animation:postStart with randomStartFrame (component_library.lua:1064):
    // the queued setState is flushed by the grid:update at api_factory.lua:101.
    for (int i = lane; i < c.n_water; i += 64) {
      const uint32_t k = philox_bounded(
          philox4x32_10((uint32_t)i, RS_ANIM_START, 0u, 0u, k0, k1), 4u);
      at(c.water_layer, c.water_cells[i]) = (uint8_t)c.s_water[k];
    }
    step_type = 0;
  } else {
    // ================= api:advance =================
    const uint64_t seed = tail->seed + (tail->episode - 1);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int step = tail->step + 1, frame = tail->frame;
    const int dirt_count0 = tail->aux_count;
    int flag0 = 0;  // GlobalData cleanedThisStep (from the previous step's flush)
    if (lane < MP_MAX_PLAYERS) {
      ax = tail->ax[lane]; ay = tail->ay[lane]; aori = tail->aori[lane];
      alive = tail->aalive[lane]; ztimer = tail->ztimer[lane]; ctimer = tail->ctimer[lane];
      achange = tail->achange[lane]; flag0 = tail->flag0[lane];
    }
    __syncthreads();
    if (lane == 0) sc->zapped_mask = 0;
    auto draw = [&](int stream, uint32_t index) {
      return philox4x32_10(index, (uint32_t)stream, (uint32_t)step, 0u, k0, k1);
    };
    // api:discreteActions (api_factory.lua:81) + the ACTION_SET lookup of
    // discrete_action_wrapper.py:97-109; Avatar:preUpdate resets the reward.
    int a_move = 0, a_turn = 0, a_zap = 0, a_clean = 0, bad = 0;
    if (is_av) {
      int a = actions[(size_t)w * P + lane];
      if (a < 0 || a >= t.nact) { a = 0; bad = 1; }
      a_move = t.action_table[a * 4 + 0]; a_turn = t.action_table[a * 4 + 1];
      a_zap = t.action_table[a * 4 + 2]; a_clean = t.action_table[a * 4 + 3];
    }
    // beam sprites of the previous frame disappear (grid:update start)
    for (int i = lane; i < HW; i += 64) { at(c.zap_layer, i) = 0; at(c.clean_layer, i) = 0; }

    // ---- BaseSimulation:update: DirtSpawner:update (clean_up/components.lua:329-340)
    if (step > c.dirt_delay) {
      const Philox4 d = draw(RS_DIRT_SPAWN, 0);
      if (philox_u53(d) < c.thr_dirt_spawn) {
        unsigned long long masks[4];
        const int n = count_sites(lane, c.n_dirt, [&](int site) {
          return at(c.dirt_wait_layer, c.dirt_cells[site]) == c.s_dirt_wait;
        }, masks);
        if (n > 0) {  // random:choice(set.toSortedList(potential))
          const int k = (int)philox_bounded(d, (uint32_t)n);
          const int site = kth_site(masks, (c.n_dirt + 63) >> 6, k);
          // first event of the flush: the DirtSpawner setState
          const int cell = c.dirt_cells[site];
          if (lane == 0 && at(c.dirt_layer, cell) == 0) {
            at(c.dirt_wait_layer, cell) = 0;
            at(c.dirt_layer, cell) = (uint8_t)c.s_dirt;
          }
        }
      }
    }
    // ---- AppleGrow:update (clean_up/components.lua:64-80): one draw per
    // potential apple; the probability depends on the dirt count only (as it
    // was when update() ran, i.e. before this frame's events).
    {
      const uint64_t thr = c.apple_thr[dirt_count0];
      for (int i = lane; i < c.n_apple; i += 64) {
        if (philox_u53(draw(RS_APPLE_GROW, (uint32_t)i)) < thr) {
          const int cell = c.apple_cells[i];
          if (at(c.apple_layer, cell) == 0) at(c.apple_layer, cell) = (uint8_t)c.s_apple;
        }
      }
    }

    // ---- updaters, priority descending (updater_registry.lua:166-173); they
    // read the pre-flush state and queue events.
    const int order_move = shuffled_order(lane, P, RS_SHUFFLE_MOVE, (uint32_t)step, k0, k1);
    const int order_zap = shuffled_order(lane, P, RS_SHUFFLE_ZAP, (uint32_t)step, k0, k1);
    const int order_resp = shuffled_order(lane, P, RS_SHUFFLE_RESPAWN, (uint32_t)step, k0, k1);
    // (the Cleaner order, RS_SHUFFLE_CLEAN, has no observable effect: beams do
    // not change state inside the flush and cleanHit carries no reward)
    bool fire_zap = false, fire_clean = false, want_respawn = false;
    if (is_av) {
      // 140 Zapper zap (avatar_library.lua:613-636)
      if (alive && c.zap_cooldown >= 0) {
        if (ztimer > 0) ztimer--;
        else if (a_zap == 1) { ztimer = c.zap_cooldown; fire_zap = true; }
      }
      // 140 Cleaner clean (clean_up/components.lua:201-224)
      if (alive && c.clean_cooldown >= 0) {
        if (ctimer > 0) ctimer--;
        else if (a_clean == 1) { ctimer = c.clean_cooldown; fire_clean = true; }
      }
      // 135 Zapper respawn: state = waitState, startFrame = framesTillRespawn
      // (avatar_library.lua:638-649)
      want_respawn = !alive && (frame - achange) >= c.respawn_frames;
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
    }
    int cleaned = 0, ate = 0;  // this frame's GlobalData flags

    // ---- flush 1: queued events in FIFO order (docs/advanced.md:43-52)
    // Avatar move (avatar_library.lua:155-203): turn, then moveRel.
    if (is_av && a_turn != 0) aori = (aori + a_turn + 4) & 3;  // off-grid pieces turn too
    const bool wants = is_av && alive && a_move != 0;
    int tx = ax, ty = ay;
    bool target_free = false;  // in bounds and no static piece on the avatar layer
    __syncthreads();           // grid writes above are visible
    if (wants) {
      const int dir = (aori + a_move - 1) & 3;
      if (step_cell(t, tx, ty, kDx[dir], kDy[dir])) {
        const int s = at(t.avatar_layer, ty * W + tx);
        target_free = s == 0 || sc->splayer[s] >= 0;  // other avatars: decided in order below
      }
    }
    const int old_cell = ay * W + ax;
    bool moved = false;
    for (int r = 0; r < P; ++r) {
      const int p = __shfl(order_move, r);
      const int ptx = __shfl(tx, p), pty = __shfl(ty, p);
      const bool pfree = __shfl((int)(wants && target_free), p) != 0;
      const bool occupied = __ballot(is_av && alive && ax == ptx && ay == pty) != 0;
      if (lane == p && pfree && !occupied) { ax = ptx; ay = pty; moved = true; }
    }
    if (moved) at(t.avatar_layer, old_cell) = 0;
    __syncthreads();
    if (moved) at(t.avatar_layer, ay * W + ax) = (uint8_t)t.alive_state[lane];
    // onContact 'avatar' enter on the destination — or, for a blocked move, on
    // the cell the avatar stays in (A3b): Edible:onEnter + Taste:consumed
    // (clean_up/components.lua:390-408,446-455); apple -> appleWait next flush.
    int ate_cell = -1;
    if (wants && at(c.apple_layer, ay * W + ax) == c.s_apple) {
      reward += c.eat_reward; ate = 1; ate_cell = ay * W + ax;
    }
    __syncthreads();

    // hitBeam (game_object.lua:246-258), footprint of Zapper:getWhoZappable
    // (avatar_library.lua:780-824): lane (b, j) = cell j of avatar b's beam.
    for (int hit = 0; hit < 2; ++hit) {
      const int nc = c.fp_n[hit];
      const int per = 64 / nc;  // beams per round
      const bool fire = hit == HIT_ZAP ? fire_zap : fire_clean;
      for (int b0 = 0; b0 < P; b0 += per) {
        const int bl = lane / nc, j = lane - bl * nc, b = b0 + bl;
        const bool lane_ok = bl < per && b < P;
        const int bs = lane_ok ? b : 0;
        const bool bfire = __shfl((int)(fire && alive), bs) != 0 && lane_ok;
        const int bx = __shfl(ax, bs), by = __shfl(ay, bs), bo = __shfl(aori, bs);
        // cell = pos + lat * right(bo) + fwd * forward(bo)
        const int lat = c.fp_lat[hit][j], fw = c.fp_fwd[hit][j];
        const int rdir = (bo + 1) & 3;
        int x = bx, y = by;
        const bool inb = step_cell(t, x, y, lat * kDx[rdir] + fw * kDx[bo],
                                   lat * kDy[rdir] + fw * kDy[bo]);
        const int cell = inb ? y * W + x : 0;
        bool blocked = false;
        int hit_player = -1;
        bool hit_dirt = false;
        if (bfire && inb) {
          for (int l = 0; l < t.L; ++l) {
            const int s = at(l, cell);
            if (s == 0) continue;
            // BeamBlocker:onHit (component_library.lua:678-685)
            if (sc->hit_block[s] & (1u << hit)) blocked = true;
            const int pl = sc->splayer[s];
            // Zapper:onHit (avatar_library.lua:652-681); on-grid => alive
            if (pl >= 0 && hit == HIT_ZAP) { hit_player = pl; blocked = true; }
            // DirtCleaning:onHit (clean_up/components.lua:141-157)
            if (hit == HIT_CLEAN && s == c.s_dirt) { hit_dirt = true; blocked = true; }
          }
        }
        // every ray stops at the first cell that is outside the map or blocks
        const unsigned long long stops = __ballot(bfire && (!inb || blocked));
        const uint32_t mine = (uint32_t)(stops >> (bl * nc)) & ((1u << nc) - 1u);
        const bool reached = bfire && inb && (mine & c.fp_pred[hit][j]) == 0;
        // A4: the beam sprite is drawn on the hit's layer, blocked cell included
        if (reached)
          at(hit == HIT_ZAP ? c.zap_layer : c.clean_layer, cell) =
              (uint8_t)(hit == HIT_ZAP ? c.s_zap_hit : c.s_clean_hit);
        const bool zhit = reached && hit_player >= 0;
        const bool dhit = reached && hit_dirt;
        if (zhit && c.remove_hit) atomicOr(&sc->zapped_mask, 1u << hit_player);
        if (dhit) mark[cell] = 1;  // dirt -> dirtWait in the next flush
        if (hit == HIT_ZAP && lane_ok) sc->victim[b][j] = (int8_t)(zhit ? hit_player : -1);
        const unsigned long long zb = __ballot(zhit), db = __ballot(dhit);
        if (lane == 0) { tail->ctr[4] += __popcll(zb); tail->ctr[5] += __popcll(db); }
        // GlobalData:setCleanedThisStep for the beam's owner
        for (int q = 0; q < per && b0 + q < P; ++q)
          if (((db >> (q * nc)) & ((1ull << nc) - 1ull)) != 0 && lane == b0 + q) cleaned = 1;
      }
      __syncthreads();
      // Zapper:onHit rewards, in the reference's event order (zap visiting
      // order, then footprint order) so that the f64 sums are bit-identical.
      if (hit == HIT_ZAP && (c.zap_penalty != 0.0 || c.zap_reward != 0.0)) {
        for (int r = 0; r < P; ++r) {
          const int owner = __shfl(order_zap, r);
          if (!(__shfl((int)fire_zap, owner) != 0)) continue;
          for (int q = 0; q < nc; ++q) {
            const int victim = sc->victim[owner][q];
            if (victim < 0) continue;
            if (lane == victim) reward += c.zap_penalty;
            if (lane == owner) reward += c.zap_reward;
          }
        }
      }
    }

    // teleportToGroup(spawnGroup, aliveState), PICK_RANDOM orientation
    // (component_library.lua:336-354).  A5: uniform over the group's pieces in
    // creation order; an occupied target fails and is retried next frame.
    int rcell = 0, rori = 0;
    bool rfree = false;
    if (want_respawn) {
      const Philox4 d = draw(RS_RESPAWN, (uint32_t)lane);
      rcell = t.spawn_cells[philox_bounded(d, (uint32_t)t.n_spawn)];
      rori = (int)(d.x3 & 3u);
      const int s = at(t.avatar_layer, rcell);
      rfree = s == 0 || sc->splayer[s] >= 0;
    }
    bool respawned = false;
    if (__any(want_respawn)) {
      for (int r = 0; r < P; ++r) {
        const int p = __shfl(order_resp, r);
        const int pc = __shfl(rcell, p);
        const bool pfree = __shfl((int)(want_respawn && rfree), p) != 0;
        const bool occupied = __ballot(is_av && alive && ay * W + ax == pc) != 0;
        if (lane == p && pfree && !occupied) {
          alive = 1; ax = pc % W; ay = pc / W; achange = frame; aori = rori;
          respawned = true;
        }
      }
      if (respawned) {
        at(t.avatar_layer, rcell) = (uint8_t)t.alive_state[lane];
        if (at(c.apple_layer, rcell) == c.s_apple) {  // placed on a live apple
          reward += c.eat_reward; ate = 1; ate_cell = rcell;
        }
      }
      const unsigned long long rb = __ballot(respawned);
      if (lane == 0) tail->ctr[6] += __popcll(rb);
    }
    // water Animation setStates: the last events of flush 1
    if (water_advance) {
      for (int i = lane; i < c.n_water; i += 64) {
        const int cell = c.water_cells[i];
        const int s = at(c.water_layer, cell);
        int k = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (s == c.s_water[q]) k = q;
        at(c.water_layer, cell) = (uint8_t)c.s_water[(k + 1) & 3];
      }
    }
    __syncthreads();

    // ---- flush 2: setStates queued by the callbacks of flush 1
    if (ate_cell >= 0) at(c.apple_layer, ate_cell) = 0;   // apple -> appleWait (off-grid)
    const uint32_t zapped = sc->zapped_mask;
    if (is_av && alive && !respawned && ((zapped >> lane) & 1u)) {  // avatar -> playerWait
      at(t.avatar_layer, ay * W + ax) = 0;
      alive = 0; achange = frame;
    }
    __syncthreads();
    int dirt_count = 0;
    {
      // sweep: apply the marked dirt -> dirtWait transitions (several beams may
      // have hit one cell) and recount
      // (RiverMonitor / DirtTracker:onStateChange, clean_up/components.lua:118-129)
      for (int ch = 0; ch * 64 < c.n_dirt; ++ch) {
        const int site = ch * 64 + lane;
        bool dirty = false;
        if (site < c.n_dirt) {
          const int cell = c.dirt_cells[site];
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
    if (lane == 0) {
      tail->step = step;
      tail->frame = frame + 1;
      tail->cont = cont;
      tail->done = !(cont && step < t.max_frames);  // api_factory.lua:107-110
      tail->aux_count = dirt_count;
      if (water_advance) tail->group_change = frame;
      tail->ctr[0]++; tail->ctr[1] += (uint32_t)P; tail->ctr[7] += __popcll(badb);
    }
    if (lane < MP_MAX_PLAYERS) { tail->flag0[lane] = (uint8_t)cleaned; tail->flag1[lane] = (uint8_t)ate; }
    __syncthreads();
    step_type = tail->done ? 2 : 1;
  }

  // ---- write the avatar registers back + outputs: "N.REWARD", "N.READY_TO_SHOOT"
  // (avatar_library.lua:737-744), NUM_OTHERS_WHO_CLEANED_THIS_STEP
  // (component_library.lua:786-803)
  if (lane < MP_MAX_PLAYERS) {
    tail->ax[lane] = (uint8_t)ax; tail->ay[lane] = (uint8_t)ay; tail->aori[lane] = (uint8_t)aori;
    tail->aalive[lane] = (uint8_t)alive; tail->ztimer[lane] = (uint8_t)ztimer;
    tail->ctimer[lane] = (uint8_t)ctimer; tail->achange[lane] = achange;
  }
  if (is_av) {
    const size_t o = (size_t)w * P + lane;
    out.reward[o] = reward;
    const double v = 1.0 - (double)ztimer / (double)c.zap_cooldown;
    out.ready[o] = alive ? (v > 0.0 ? v : 0.0) : 0.0;
    out.aux0[o] = aux0;
    out.position[o * 2 + 0] = ax;
    out.position[o * 2 + 1] = ay;
    out.orientation[o] = aori;
  }
  {
    // COLLECTIVE_REWARD = sum over players in index order (collective_reward_wrapper.py:49)
    double sum = 0.0;
    for (int p = 0; p < P; ++p) sum += __shfl(reward, p);
    if (lane == 0) {
      out.collective[w] = sum;
      out.step_type[w] = step_type;
      out.discount[w] = step_type == 1 ? 1.0 : 0.0;
      tail->reward_fx += (uint32_t)(int32_t)(sum * 1024.0);
    }
  }
  __syncthreads();
  for (int i = lane; i < nvec; i += 64)
    reinterpret_cast<uint4*>(gw)[i] = reinterpret_cast<const uint4*>(smem)[i];
}

}  // namespace

void launch_step_clean_up(const DevTables& t, const CleanUpTables& c,
                          uint8_t* state, int num_worlds, const int32_t* actions,
                          const uint8_t* reset_mask, int mode, int auto_reset,
                          const StepOutputs& out, hipStream_t stream) {
  const size_t lds = (size_t)t.world_stride + sizeof(Scratch) + (size_t)((t.H * t.W + 15) & ~15);
  hipLaunchKernelGGL(k_step_clean_up, dim3(num_worlds), dim3(64), lds, stream, t,
                     c, state, actions, reset_mask, mode, auto_reset, out);
}
