// step_commons.hip — one environment step (or episode start) of N
// commons_harvest worlds, one wavefront per world (shape: step_clean_up.hip).
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
#include "step_common.h"

namespace {

using namespace stepk;

enum { HIT_ZAP = 0 };

__global__ __launch_bounds__(64) void k_step_commons(
    DevTables t, CommonsTables c, uint8_t* __restrict__ state,
    const int32_t* __restrict__ actions, const uint8_t* __restrict__ reset_mask,
    int mode, int auto_reset, StepOutputs out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  uint8_t* gw = state + (size_t)w * t.world_stride;
  const Action act = fetch_action(t, actions, mode, w, lane);
  load_world(t, smem, gw, lane);
  Scratch* sc = reinterpret_cast<Scratch*>(smem + t.world_stride);
  uint8_t* grid = smem;
  WorldTail* tail = reinterpret_cast<WorldTail*>(smem + t.grid_pad);
  const int P = t.P, HW = t.H * t.W, W = t.W, H = t.H;
  const bool is_av = lane < P;
  auto at = [&](int layer, int cell) -> uint8_t& { return grid[layer * HW + cell]; };

  const int what = dispatch(t, tail, lane, w, reset_mask, mode, auto_reset, out);
  if (what == 0) return;

  Av a;
  int step_type;

  if (what == 1) {
    // ---- api:start (api_factory.lua:85-102); seed + #earlier resets (builder.py:177-181)
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
      tail->aux_count = c.n_apple;  // every apple starts live
      tail->group_change = 0;
      tail->ctr[2]++;
    }
    if (lane < MP_MAX_PLAYERS) { tail->flag0[lane] = 0; tail->flag1[lane] = 0; }
    __syncthreads();
    apply_map_choices(t, grid, lane, k0, k1);
    spawn_avatars(t, grid, lane, k0, k1, a);
    if (lane < P) push_event(sc, MP_EVENT_AVATAR_STARTED, 0, 0);
    step_type = 0;
  } else {
    // ================= api:advance =================
    const uint64_t seed = tail->seed + (tail->episode - 1);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int step = tail->step + 1, frame = tail->frame;
    load_avatars(tail, lane, a);
    __syncthreads();
    if (lane == 0) sc->zapped_mask = 0;
    auto draw = [&](int stream, uint32_t index) {
      return philox4x32_10(index, (uint32_t)stream, (uint32_t)step, 0u, k0, k1);
    };
    const int a_move = act.move, a_turn = act.turn, a_zap = act.fire0, bad = act.bad;

    // ---- per waiting apple (one lane each, up to 4 rounds):
    //  * DensityRegrow sprout updater (priority 10): decided on the state the
    //    piece has NOW (set by the previous frame), A12: one draw per piece;
    //  * DensityRegrow:update -> _updateWaitState (components.lua:161-193):
    //    k = live apples in the disc, as of the end of the previous frame.
    uint32_t sprout_bits = 0, wait_bits = 0;
    int new_k[4] = {0, 0, 0, 0};
    for (int r = 0; r * 64 < c.n_apple && r < 4; ++r) {
      const int i = r * 64 + lane;
      if (i >= c.n_apple) continue;
      const int cell = c.apple_cells[i];
      const int ws = at(c.wait_layer, cell);
      if (ws == 0 || at(c.live_layer, cell) == c.s_apple) continue;  // live
      wait_bits |= 1u << r;
      int old_k = -1;
      for (int k = 0; k < c.nk; ++k) if (ws == c.s_wait_k[k]) old_k = k;
      if (old_k >= 0 && philox_u53(draw(RS_REGROW, (uint32_t)i)) < c.thr[old_k])
        sprout_bits |= 1u << r;
      const int x0 = cell % W, y0 = cell / W;
      int n = 0;
      for (int d = 0; d < c.ndisc; ++d) {
        int x = x0 + c.disc[2 * d], y = y0 + c.disc[2 * d + 1];
        if (t.topology == 1) { x = ((x % W) + W) % W; y = ((y % H) + H) % H; }
        else if (x < 0 || x >= W || y < 0 || y >= H) continue;
        n += at(c.live_layer, y * W + x) == c.s_apple;
      }
      new_k[r] = n;
    }
    // beam sprites of the previous frame disappear (grid:update start)
    for (int i = lane; i < HW; i += 64) at(c.zap.layer, i) = 0;
    __syncthreads();
    // first events of the flush: setState(appleWait_k) and the grass under it
    for (int r = 0; r < 4; ++r) {
      if (!((wait_bits >> r) & 1u)) continue;
      const int cell = c.apple_cells[r * 64 + lane];
      at(c.wait_layer, cell) = (uint8_t)c.s_wait_k[new_k[r]];
      const int g = at(c.grass_layer, cell);
      if (g == c.s_grass || g == c.s_dess)
        at(c.grass_layer, cell) = (uint8_t)(new_k[r] == 0 ? c.s_dess : c.s_grass);
    }

    // ---- updaters (pre-flush state)
    int orders[4];
    shuffled_orders(lane, P, {RS_SHUFFLE_MOVE, RS_SHUFFLE_ZAP, RS_SHUFFLE_RESPAWN, 0}, 3,
                    (uint32_t)step, k0, k1, orders);
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

    // ---- flush 1
    const bool wants = resolve_moves(t, grid, sc, lane, a, a_move, a_turn, order_move);
    // Edible:onEnter (component_library.lua:990-1004); apple -> appleWait next flush
    int ate_cell = -1;
    if (wants && at(c.live_layer, a.y * W + a.x) == c.s_apple) {
      a.reward += c.eat_reward; ate_cell = a.y * W + a.x;
      push_event(sc, MP_EVENT_EDIBLE_CONSUMED, lane + 1, 0);
    }
    __syncthreads();
    fire_beams(t, grid, sc, tail, lane, a, fire_zap, beam_lane(c.zap.shape, lane), c.zap.hit, true,
               c.zap.layer, c.zap.s_hit, c.zap.remove_hit != 0,
               [](int, int) { return 0; },
               [](int, int, int, bool, int, bool) {});
    zap_rewards(t, sc, lane, a, fire_zap, order_zap, c.zap.shape.n, c.zap.penalty,
                c.zap.reward);
    const int rcell = resolve_respawns(t, grid, sc, tail, lane, a, want_respawn, order_resp,
                                       (uint32_t)step, frame, k0, k1);
    if (rcell >= 0 && at(c.live_layer, rcell) == c.s_apple) {
      a.reward += c.eat_reward; ate_cell = rcell;
      push_event(sc, MP_EVENT_EDIBLE_CONSUMED, lane + 1, 0);
    }
    __syncthreads();
    // sprouts: setState(apple), the last events of flush 1 (canRegrowIfOccupied)
    for (int r = 0; r < 4; ++r) {
      if (!((sprout_bits >> r) & 1u)) continue;
      const int cell = c.apple_cells[r * 64 + lane];
      if (at(c.live_layer, cell) == 0) {
        at(c.wait_layer, cell) = 0;
        at(c.live_layer, cell) = (uint8_t)c.s_apple;
      }
    }
    __syncthreads();

    // ---- flush 2
    if (ate_cell >= 0 && at(c.wait_layer, ate_cell) == 0) {  // apple -> appleWait
      at(c.live_layer, ate_cell) = 0;
      at(c.wait_layer, ate_cell) = (uint8_t)c.s_wait;
    }
    apply_zapped(t, grid, sc, lane, a, rcell >= 0, frame);
    __syncthreads();
    int live = 0;
    for (int r = 0; r * 64 < c.n_apple; ++r) {
      const int i = r * 64 + lane;
      live += __popcll(__ballot(i < c.n_apple &&
                                at(c.live_layer, c.apple_cells[i < c.n_apple ? i : 0]) == c.s_apple));
    }
    const unsigned long long badb = __ballot(bad != 0);
    if (lane == 0) {
      tail->step = step;
      tail->frame = frame + 1;
      tail->cont = cont;
      tail->done = !(cont && step < t.max_frames);
      tail->aux_count = live;
      tail->ctr[0]++; tail->ctr[1] += (uint32_t)P; tail->ctr[7] += __popcll(badb);
    }
    __syncthreads();
    step_type = tail->done ? 2 : 1;
  }
  finish(t, smem, gw, tail, lane, w, a, 0.0, c.zap.cooldown, step_type, out);
}

}  // namespace

void launch_step_commons(const DevTables& t, const CommonsTables& c,
                         uint8_t* state, int num_worlds, const int32_t* actions,
                         const uint8_t* reset_mask, int mode, int auto_reset,
                         const StepOutputs& out, hipStream_t stream) {
  hipLaunchKernelGGL(k_step_commons, dim3(num_worlds), dim3(64), stepk::lds_bytes(t),
                     stream, t, c, state, actions, reset_mask, mode, auto_reset, out);
}
