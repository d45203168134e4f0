// step_kernels.hip — the stand-alone step kernels: one wavefront per world, four
// worlds per workgroup (they share the read-only LDS tables).  They run the same wave-level step functions
// (step_<substrate>.h) as the fused step + render kernels of frame.hip, and are
// what an engine launches when no RGB observation is bound to a step (and for
// mp_reset).  Reference path replaced: api:advance / api:start
// (lua/modules/api_factory.lua:85-111) — see step_clean_up.h.
#include "../../include/mp_pack.h"
#include "step_clean_up.h"
#include "step_coins.h"
#include "step_commons.h"
#include "step_coop.h"
#include "step_gift.h"
#include "step_mushroom.h"
#include "step_cook.h"
#include "step_matrix.h"
#include "step_territory.h"

namespace {

using namespace stepk;

constexpr int kWorldsPerGroup = 4;   // waves of a workgroup; they share the LDS tables

template <class Tables, class Sites>
__device__ inline void run_one_world(const DevTables& t, const Tables& c, const StepArgs& args,
                                     int extra) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int w = blockIdx.x * kWorldsPerGroup + wave;
  // LDS: [tables][wave 0: record, scratch, marks, extra][wave 1: ...]...
  uint8_t* tables = smem;
  const int per_world = t.world_stride + scratch_bytes(t) + extra;
  uint8_t* mine = smem + tables_bytes(t) + wave * per_world;
  const bool live = w < args.num_worlds;
#ifdef MP_STEP_TIMING
  const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
  World wd = make_world(t, mine, tables, mine + t.world_stride, args.state, live ? w : 0, lane);
  wd.next_orders = args.next_orders;
  // every global read of the step is issued here, before the first wait: the
  // action id, the site lists, the record, the tables — one trip to memory
  int act_id = 0;
  Sites sites = Sites();
  if (live) {
    act_id = fetch_action_id(t, args.actions, args.mode, w, lane);
    sites = load_sites(c, lane);
    load_record(t, wd.rec, wd.gw, lane);
  }
  load_tables(t, tables, (int)threadIdx.x, kWorldsPerGroup * 64);
  clear_marks(t, wd.mark, lane);
  begin_step(wd.sc, lane);
  __syncthreads();   // the tables are the one thing the waves of a group share
  if (!live) return;
#ifdef MP_STEP_TIMING
  if (lane == 0 && (w == 7 || w == 2000) && args.mode == STEP_MODE_STEP)
    printf("w %d: entry -> record in LDS %llu cycles\n", w, __builtin_readcyclecounter() - t_entry);
#endif
  const Action act = lookup_action(t, wd, act_id, args.mode);
  init_extra(t, c, wd.extra, lane);
  step_world(t, c, sites, wd, act, args);
}

__global__ __launch_bounds__(kWorldsPerGroup * 64) void k_step_clean_up(DevTables t, CleanUpTables c, StepArgs args) {
  run_one_world<CleanUpTables, CleanUpSites>(t, c, args, 0);
}
__global__ __launch_bounds__(kWorldsPerGroup * 64) void k_step_commons(DevTables t, CommonsTables c, StepArgs args) {
  run_one_world<CommonsTables, CommonsSites>(t, c, args, 0);
}
__global__ __launch_bounds__(kWorldsPerGroup * 64) void k_step_coins(DevTables t, CoinsTables c, StepArgs args) {
  run_one_world<CoinsTables, CoinsSites>(t, c, args, 0);
}
__global__ __launch_bounds__(kWorldsPerGroup * 64) void k_step_coop(DevTables t, CoopTables c, StepArgs args) {
  run_one_world<CoopTables, CoopSites>(t, c, args, 0);
}
__global__ __launch_bounds__(kWorldsPerGroup * 64) void k_step_gift(DevTables t, GiftTables c, StepArgs args) {
  run_one_world<GiftTables, GiftSites>(t, c, args, 0);
}
__global__ __launch_bounds__(kWorldsPerGroup * 64) void k_step_cook(DevTables t, CookTables c, StepArgs args) {
  run_one_world<CookTables, CookSites>(t, c, args, 0);
}
__global__ __launch_bounds__(kWorldsPerGroup * 64) void k_step_mushroom(DevTables t, MushroomTables c, StepArgs args) {
  run_one_world<MushroomTables, MushroomSites>(t, c, args, extra_bytes(c));
}
__global__ __launch_bounds__(kWorldsPerGroup * 64) void k_step_matrix(DevTables t, MatrixTables c, StepArgs args) {
  run_one_world<MatrixTables, MatrixSites>(t, c, args, 0);
}
__global__ __launch_bounds__(kWorldsPerGroup * 64) void k_step_territory(DevTables t, TerritoryTables c, StepArgs args) {
  run_one_world<TerritoryTables, TerritorySites>(t, c, args, extra_bytes(c));
}

// "N.LAYER" (avatar_library.lua:246-257): the player's layer view with
// orientation 'N', as sprite ids (A17; oracle/render.c: orc_layer_view).  A debug
// observation read straight from the records in HBM: one thread per (world,
// player, window cell).
__global__ void k_layer_view(DevTables t, const uint8_t* __restrict__ state,
                             int32_t* __restrict__ out, int num_worlds) {
  const int VW = t.vl + t.vr + 1, VH = t.vf + t.vb + 1, HW = t.H * t.W;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)num_worlds * t.P * VH * VW) return;
  const int vx = (int)(i % VW), vy = (int)((i / VW) % VH);
  const int p = (int)((i / (VW * VH)) % t.P), w = (int)(i / ((long long)VW * VH * t.P));
  const uint8_t* rec = state + (size_t)w * t.world_stride;
  const WorldTail* tail = reinterpret_cast<const WorldTail*>(rec + t.grid_pad);
  const int32_t* remap = t.view_sprite_map + (size_t)p * t.nsprites;
  int32_t* dst = out + (size_t)i * t.L;
  int x = tail->ax[p] + (vx - t.vl), y = tail->ay[p] + (vy - t.vf);
  bool in = true;
  if (t.topology == 1) { x = ((x % t.W) + t.W) % t.W; y = ((y % t.H) + t.H) % t.H; }
  else in = x >= 0 && x < t.W && y >= 0 && y < t.H;
  if (!in || !tail->aalive[p]) {   // A6: an off-grid viewer sees only OutOfBounds
    for (int l = 0; l < t.L; ++l) dst[l] = 1 + remap[0];
    return;
  }
  for (int l = 0; l < t.L; ++l) {
    const int s = rec[l * HW + y * t.W + x];
    const int sprite = s ? t.state_sprite[s] : -1;
    dst[l] = sprite >= 0 ? 1 + remap[sprite] : 0;
  }
}

}  // namespace

void launch_layer_view(const DevTables& t, const uint8_t* state, int32_t* out, int num_worlds,
                       hipStream_t stream) {
  const long long n = (long long)num_worlds * t.P * (t.vl + t.vr + 1) * (t.vf + t.vb + 1);
  hipLaunchKernelGGL(k_layer_view, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t,
                     state, out, num_worlds);
}

void launch_step(const DevTables& t, const SubstrateTables& s, const stepk::StepArgs& args,
                 hipStream_t stream) {
  const int extra = s.substrate == MPK_SUBSTRATE_TERRITORY ? stepk::extra_bytes(s.tr)
                    : s.substrate == MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS ? stepk::extra_bytes(s.em) : 0;
  const size_t lds = (size_t)stepk::tables_bytes(t) +
                     (size_t)kWorldsPerGroup * (t.world_stride + stepk::scratch_bytes(t) + extra);
  const dim3 grid((args.num_worlds + kWorldsPerGroup - 1) / kWorldsPerGroup), block(kWorldsPerGroup * 64);
  switch (s.substrate) {
    case MPK_SUBSTRATE_CLEAN_UP:
      hipLaunchKernelGGL(k_step_clean_up, grid, block, lds, stream, t, s.cu, args);
      break;
    case MPK_SUBSTRATE_COMMONS_HARVEST:
      hipLaunchKernelGGL(k_step_commons, grid, block, lds, stream, t, s.ch, args);
      break;
    case MPK_SUBSTRATE_COINS:
      hipLaunchKernelGGL(k_step_coins, grid, block, lds, stream, t, s.co, args);
      break;
    case MPK_SUBSTRATE_TERRITORY:
      hipLaunchKernelGGL(k_step_territory, grid, block, lds, stream, t, s.tr, args);
      break;
    case MPK_SUBSTRATE_THE_MATRIX:
      hipLaunchKernelGGL(k_step_matrix, grid, block, lds, stream, t, s.mx, args);
      break;
    case MPK_SUBSTRATE_COOP_MINING:
      hipLaunchKernelGGL(k_step_coop, grid, block, lds, stream, t, s.cm, args);
      break;
    case MPK_SUBSTRATE_GIFT_REFINEMENTS:
      hipLaunchKernelGGL(k_step_gift, grid, block, lds, stream, t, s.gr, args);
      break;
    case MPK_SUBSTRATE_COLLABORATIVE_COOKING:
      hipLaunchKernelGGL(k_step_cook, grid, block, lds, stream, t, s.cc, args);
      break;
    case MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS:
      hipLaunchKernelGGL(k_step_mushroom, grid, block, lds, stream, t, s.em, args);
      break;
  }
}
