// step_coins.hip — one environment step (or episode start) of N coins worlds,
// one wavefront per world (shape: step_clean_up.hip).
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
#include "step_common.h"

namespace {

using namespace stepk;

__global__ __launch_bounds__(64) void k_step_coins(
    DevTables t, CoinsTables c, uint8_t* __restrict__ state,
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
  const int P = t.P, HW = t.H * t.W, W = t.W;
  const bool is_av = lane < P;
  auto at = [&](int layer, int cell) -> uint8_t& { return grid[layer * HW + cell]; };

  const int what = dispatch(t, tail, lane, w, reset_mask, mode, auto_reset, out);
  if (what == 0) return;

  Av a;
  double aux0 = 0.0;
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
      tail->aux_count = 0;  // every coin starts in coinWait
      tail->group_change = 0;
      tail->ctr[2]++;
    }
    if (lane < MP_MAX_PLAYERS) { tail->flag0[lane] = 0; tail->flag1[lane] = 0; }
    __syncthreads();
    apply_map_choices(t, grid, lane, k0, k1);
    spawn_avatars(t, grid, lane, k0, k1, a);
    if (is_av) push_event(sc, MP_EVENT_AVATAR_STARTED, 0, 0);
    step_type = 0;
  } else {
    // ================= api:advance =================
    const uint64_t seed = tail->seed + (tail->episode - 1);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int step = tail->step + 1, frame = tail->frame;
    load_avatars(tail, lane, a);
    __syncthreads();
    auto draw = [&](int stream, uint32_t index) {
      return philox4x32_10(index, (uint32_t)stream, (uint32_t)step, 0u, k0, k1);
    };
    // ---- updaters (pre-flush state)
    int orders[4];
    shuffled_orders(lane, P, {RS_SHUFFLE_MOVE, 0, 0, 0}, 1, (uint32_t)step, k0, k1, orders);
    const int order_move = orders[0];
    int cont = tail->cont;  // StochasticIntervalEpisodeEnding: _t == step + 1
    if (frame >= c.ee_min_frames && (step + 1) % c.ee_interval == 0)
      if (philox_u53(draw(RS_EPISODE_END, 0)) < c.thr_ee) cont = 0;
    // ChoiceCoinRegrow: a draw per waiting coin (A12), then random:choice of the
    // colour; the setState is queued behind the moves of this flush
    constexpr int kPerLane = 8;   // mp_create admits at most 512 coins
    uint32_t regrow = 0;          // 2 bits per coin of this lane: 0 none, 1 + colour
#pragma unroll
    for (int r = 0; r < kPerLane; ++r) {
      const int i = r * 64 + lane;
      if (i >= c.n_coin) break;
      if (at(c.wait_layer, c.coin_cells[i]) != c.s_wait) continue;
      if (philox_u53(draw(RS_REGROW, (uint32_t)i)) >= c.thr_regrow) continue;
      regrow |= (1u + philox_bounded(draw(RS_COIN_CHOICE, (uint32_t)i), 2u)) << (2 * r);
    }

    // ---- flush 1: the moves, in visiting order
    const bool wants = resolve_moves(t, grid, sc, lane, a, act.move, act.turn, order_move);
    // Coin:onEnter on the cell the avatar is in now (A3b: a blocked move
    // re-enters its own cell)
    int got = -1, got_cell = -1;   // colour of the coin this avatar collected
    if (wants) {
      const int s = at(c.coin_layer, a.y * W + a.x);
      if (s == c.s_coin[0] || s == c.s_coin[1]) { got = s == c.s_coin[1]; got_cell = a.y * W + a.x; }
    }
    for (int p = 0; p < P; ++p) {   // rewards: collector and everyone else
      const int gp = __shfl(got, p);
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
    __syncthreads();
    // the regrown coins appear (the last events of flush 1)
#pragma unroll
    for (int r = 0; r < kPerLane; ++r) {
      const uint32_t g = (regrow >> (2 * r)) & 3u;
      if (!g) continue;
      const int cell = c.coin_cells[r * 64 + lane];
      at(c.wait_layer, cell) = 0;
      at(c.coin_layer, cell) = (uint8_t)c.s_coin[g - 1u];
    }
    __syncthreads();
    // ---- flush 2: collected coins go to coinWait
    if (got_cell >= 0) {
      at(c.coin_layer, got_cell) = 0;
      at(c.wait_layer, got_cell) = (uint8_t)c.s_wait;
    }
    __syncthreads();
    int live = 0;
    for (int r = 0; r * 64 < c.n_coin; ++r) {
      const int i = r * 64 + lane;
      live += __popcll(__ballot(i < c.n_coin &&
                                at(c.coin_layer, c.coin_cells[i < c.n_coin ? i : 0]) != 0));
    }
    const unsigned long long badb = __ballot(act.bad != 0);
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
  // "N.MISMATCHED_COIN_COLLECTED_BY_PARTNER" is the substrate metric; no Zapper:
  // READY_TO_SHOOT stays 1 (timer 0, cooldown 1)
  finish(t, smem, gw, tail, lane, w, a, aux0, 1, step_type, out);
}

}  // namespace

void launch_step_coins(const DevTables& t, const CoinsTables& c,
                       uint8_t* state, int num_worlds, const int32_t* actions,
                       const uint8_t* reset_mask, int mode, int auto_reset,
                       const StepOutputs& out, hipStream_t stream) {
  hipLaunchKernelGGL(k_step_coins, dim3(num_worlds), dim3(64), stepk::lds_bytes(t),
                     stream, t, c, state, actions, reset_mask, mode, auto_reset, out);
}
