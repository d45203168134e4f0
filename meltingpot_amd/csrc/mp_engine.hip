// mp_engine.hip — C ABI of libmp_engine.so (declared in include/mp_engine.h).
//
// Host side of the boundary that replaces dmlab2d.Lab2d / dmlab2d.Environment
// (reference: meltingpot/utils/substrates/builder.py:179-187,
// wrappers/base.py:38-84).  No CPU execution path exists here: every call that
// would compute needs a HIP device and fails loudly without one.
#include <chrono>
#include <functional>
#include <map>
#include <mutex>
#include "../../include/mp_engine.h"

#include <stdarg.h>
#include <sys/types.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/mp_pack.h"
#include "step_common.h"
#include "step_matrix.h"   // MxPlayer (record layout)

void launch_step(const DevTables& t, const SubstrateTables& s, const stepk::StepArgs& args,
                 hipStream_t stream);
void launch_layer_view(const DevTables& t, const uint8_t* state, int32_t* out, int num_worlds,
                       hipStream_t stream);

// frame.hip
constexpr int kFaultWords = 64 + 4 * 16 * 64 * 2;   // fault words + the timeline build's log
// views: 0 = per-agent RGB, 1 = WORLD.RGB, 2 = both in one launch
FramePlan plan_frame(const DevTables& t, const SubstrateTables& s, int num_worlds,
                     bool with_step, int views, int num_cus, const MpDevOptions* dev);
int frame_lds_bytes(const DevTables& t, const FramePlan& p);
int render_blob_bytes(const DevTables& t);
int prepare_frame();
void build_render_blob(const DevTables& t, const uint8_t* images, const uint16_t* img_slot,
                       const uint32_t* pair_table, const int32_t* state_sprite,
                       const int8_t* state_player, const int32_t* view_sprite_map,
                       const uint8_t* sprite_flags8, const int32_t* state_orient,
                       uint8_t* blob);
uint32_t render_visible_layers(const DevTables& t, const uint8_t* blob, const int32_t* state_layer);
void launch_frame(const DevTables& t, const SubstrateTables* s, const stepk::StepArgs& args,
                  uint8_t* out_a, uint8_t* out_w, const FramePlan& p, hipStream_t stream);

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                       \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess)                                                   \
      return fail(MP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

template <class T> struct MpkType;
template <> struct MpkType<uint8_t> { static constexpr uint32_t code = MPK_U8; };
template <> struct MpkType<char> { static constexpr uint32_t code = MPK_U8; };
template <> struct MpkType<int32_t> { static constexpr uint32_t code = MPK_I32; };
template <> struct MpkType<double> { static constexpr uint32_t code = MPK_F64; };
template <> struct MpkType<uint64_t> { static constexpr uint32_t code = MPK_U64; };
template <> struct MpkType<uint32_t> { static constexpr uint32_t code = MPK_U32; };

// Table `name` of element type T (NULL if absent or of another type); payloads
// are 16-byte aligned (mpk_validate), so int4 / uint4 reads of them are legal.
template <class T>
const T* table(const void* pack, const char* name, uint64_t* count = nullptr) {
  return static_cast<const T*>(mpk_require(pack, name, MpkType<T>::code, 0, count));
}

// ... with at least `min_count` elements.
template <class T>
const T* table_n(const void* pack, const char* name, uint64_t min_count) {
  return static_cast<const T*>(mpk_require(pack, name, MpkType<T>::code, min_count, nullptr));
}

// Every value of `v[0, n)` lies in [lo, hi).
bool in_range(const int32_t* v, uint64_t n, int64_t lo, int64_t hi) {
  for (uint64_t i = 0; i < n; ++i)
    if (v[i] < lo || v[i] >= hi) return false;
  return true;
}

}  // namespace

struct MpEngine {
  int device = 0;
  int N = 0;
  int auto_reset = 0;
  hipStream_t stream = nullptr;
  int substrate = 0;
  DevTables t{};
  SubstrateTables sub{};
  CleanUpTables& cu = sub.cu;
  CommonsTables& ch = sub.ch;
  TerritoryTables& tr = sub.tr;
  CoinsTables& co = sub.co;
  MatrixTables& mx = sub.mx;
  CoopTables& cm = sub.cm;
  GiftTables& gr = sub.gr;
  CookTables& cc = sub.cc;
  MushroomTables& em = sub.em;
  // resource / token classes of "N.INVENTORY" (0: the level has no such observation)
  int inventory_types() const {
    return substrate == MPK_SUBSTRATE_THE_MATRIX ? sub.mx.R
           : substrate == MPK_SUBSTRATE_GIFT_REFINEMENTS ? sub.gr.ntypes : 0;
  }
  std::vector<uint8_t> pack;       // host copy
  uint8_t* d_pack = nullptr;       // device copy of the pack
  uint8_t* d_extra = nullptr;      // derived tables (opaque flags, state->player)
  uint8_t* d_stepblob = nullptr;   // the step kernels' LDS tables (step_common.h)
  uint8_t* d_debug = nullptr;      // engine-owned debug observations (MpConfig.debug_observations)
  uint32_t* h_fault = nullptr;     // DevTables::fault: pinned, device-mapped host memory [64]
  uint8_t* d_state = nullptr;      // [N][world_stride]
  uint8_t* d_scalars = nullptr;    // engine-owned scalar outputs
  size_t scalars_bytes = 0, debug_bytes = 0;   // of d_scalars / d_debug (mp_tune saves them)
  StepOutputs own{};               // views into d_scalars
  void* bound[MP_OBS_KINDS] = {};
  // The rollout ring (mp_bind_output_ring): submission t since the ring was bound writes
  // slot t % ring_slots of every ring-bound kind.  Between submissions bound[kind] of a ring
  // kind is the slot written LAST (what mp_observe reads); before anything was submitted,
  // slot 0.
  struct RingKind { uint8_t* base = nullptr; uint64_t stride = 0; };
  RingKind ring[MP_OBS_KINDS];
  int ring_slots = 0;              // 0: no kind is ring-bound
  uint64_t ring_cursor = 0;        // submissions since the ring was bound
  bool ring_hold = false;          // mp_tune: submissions stay on the slot it pointed at
  std::vector<FramePlan> ring_plan[3];   // [views]: the plan mp_tune kept for each slot (empty: plan[1][views])
  void point_ring(int slot, bool pixels_only = false) {
    for (int k = 0; k < MP_OBS_KINDS; ++k)
      if (ring[k].base && (!pixels_only || k == MP_OBS_RGB || k == MP_OBS_WORLD_RGB))
        bound[k] = ring[k].base + (uint64_t)slot * ring[k].stride;
  }
  bool ring_has_pixels() const { return ring[MP_OBS_RGB].base || ring[MP_OBS_WORLD_RGB].base; }
  int32_t* d_actions = nullptr;    // staging for mp_step_host
  int32_t* d_fields = nullptr;     // staging for mp_step_fields_host
  uint8_t* d_mask = nullptr;       // staging for mp_reset
  uint64_t* d_seeds = nullptr;
  unsigned long long* d_ctr = nullptr;  // mp_counters accumulator
  // mp_step_host: ring of pinned, device-mapped action buffers
  static constexpr int kHostSlots = 4;
  int32_t* h_actions[kHostSlots] = {};
  hipEvent_t h_copied[kHostSlots] = {};
  uint64_t host_steps = 0;
  FramePlan plan[2][3] = {};       // frame kernel geometry [drawing only, stepping + drawing][agents, world view, both]
  int frame_launches = 0;          // parity of DevTables::claim's counters (FramePlan::parity)
  uint32_t* d_claim = nullptr;     // DevTables::claim
  int num_cus = 0;
  int next_orders = 1;             // StepArgs::next_orders (MpDevOptions.no_next_orders turns it off)
  bool has_dev = false;            // MpConfig.dev given: the plans are the caller's, mp_tune keeps them
  bool touched = false;            // reset / stepped / restored since creation (mp_tune: may it really step?)
  int unfused = 0;                 // MpConfig.unfused: 0 the engine's choice, 1 two launches, 2 one
  // The engine's choice (MpConfig.unfused = 0): one launch, always.  (Round 2 drew
  // views under 64 KB a world — the two-player games — in a second launch: a CU
  // then has 64 worlds to step for 2 us of drawing each, and with 4-8 feeders the
  // fused form lost, 160 vs 113 us.  With batches of 8, 8 feeders (coins: 4) and the
  // round-3 kernel it wins: prisoners_dilemma repeated 102 vs 114 us, coins 156 vs
  // 190; plan_frame, tools/gpu_small_views.sh.)
  bool fuse(bool world_view) const {
    (void)world_view;
    return unfused != 1;
  }
  uint8_t* d_atlas = nullptr;      // de-duplicated atlas + image slots
  int nhits = 0;

  template <class T>
  const T* dev(const void* host_table) const {
    return reinterpret_cast<const T*>(
        d_pack + (static_cast<const uint8_t*>(host_table) - pack.data()));
  }
  StepOutputs outputs() const {
    StepOutputs o = own;
    if (bound[MP_OBS_REWARD]) o.reward = (double*)bound[MP_OBS_REWARD];
    if (bound[MP_OBS_READY_TO_SHOOT]) o.ready = (double*)bound[MP_OBS_READY_TO_SHOOT];
    if (bound[MP_OBS_AUX0]) o.aux0 = (double*)bound[MP_OBS_AUX0];
    if (bound[MP_OBS_STEP_TYPE]) o.step_type = (int32_t*)bound[MP_OBS_STEP_TYPE];
    if (bound[MP_OBS_DISCOUNT]) o.discount = (double*)bound[MP_OBS_DISCOUNT];
    if (bound[MP_OBS_COLLECTIVE_REWARD]) o.collective = (double*)bound[MP_OBS_COLLECTIVE_REWARD];
    if (bound[MP_OBS_POSITION]) o.position = (int32_t*)bound[MP_OBS_POSITION];
    if (bound[MP_OBS_ORIENTATION]) o.orientation = (int32_t*)bound[MP_OBS_ORIENTATION];
    if (bound[MP_OBS_EVENTS]) o.events = (int32_t*)bound[MP_OBS_EVENTS];
    for (int k = 0; k < 4; ++k)
      if (bound[MP_OBS_AUX1 + k]) o.dbg[k] = (double*)bound[MP_OBS_AUX1 + k];
    if (bound[MP_OBS_ZAP_MATRIX]) o.zap_matrix = (double*)bound[MP_OBS_ZAP_MATRIX];
    if (bound[MP_OBS_INVENTORY]) o.inventory = (double*)bound[MP_OBS_INVENTORY];
    if (bound[MP_OBS_INTERACTION_INVENTORIES])
      o.interaction = (double*)bound[MP_OBS_INTERACTION_INVENTORIES];
    if (bound[MP_OBS_MATRIX_CUMULANTS]) o.cumulants = (double*)bound[MP_OBS_MATRIX_CUMULANTS];
    if (bound[MP_OBS_INTERACTION_REWARDS])
      o.interaction_rewards = (double*)bound[MP_OBS_INTERACTION_REWARDS];
    return o;
  }
};

namespace {

// mp_tune's probe actions: uniform over the ACTION_SET, a hash of the index (what a
// random policy — and bench.py — sends; NOOP steps cost 5 % less than real ones)
// mp_box_fill's store loop: the frame launch's store FORM (persistent workgroups, whole spans
// per wave from an LDS ticket counter, 16-byte lane-contiguous non-temporal stores: 1 KiB per
// wave instruction) with nothing but the stores — what the memory system takes from this
// write order on this buffer.  order 0: workgroup g owns bytes [g * own, (g + 1) * own) and
// walks them in spans of `span`; order 1: one chip-wide front, turn t of workgroup g is span
// t * G + g of the whole view.
__global__ __launch_bounds__(1024) void k_box_fill(uint8_t* out, uint64_t bytes, uint64_t own,
                                                   uint32_t span, int order) {
  __shared__ uint32_t next;
  if (threadIdx.x == 0) next = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t g = blockIdx.x, G = gridDim.x;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&next, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    uint64_t begin, end;
    if (order == 0) {
      uint64_t lim = (g + 1) * own;
      if (lim > bytes) lim = bytes;
      begin = g * own + (uint64_t)t * span;
      if (begin >= lim) break;
      end = begin + span < lim ? begin + span : lim;
    } else {
      begin = ((uint64_t)t * G + g) * span;
      if (begin >= bytes) break;
      end = begin + span < bytes ? begin + span : bytes;
    }
    const uint64_t sp = reinterpret_cast<uint64_t>(out + begin);
    uint8_t* base = reinterpret_cast<uint8_t*>(
        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sp));
    const uint32_t n = (uint32_t)(end - begin);
    for (uint32_t off = lane * 16u; off < n; off += 1024u)
      asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(off), "v"(u32x4{t, off, 2u, 3u}), "s"(base));
  }
}

__global__ void k_probe_actions(int32_t* actions, int n, int nact, uint32_t salt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = (uint32_t)i * 2654435761u + salt;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  actions[i] = (int32_t)(x % (uint32_t)nact);
}

__global__ void k_set_seeds(uint8_t* state, int stride, int grid_pad, int n,
                            const uint64_t* seeds, const uint8_t* mask) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n || (mask && !mask[w])) return;
  WorldTail* tail = reinterpret_cast<WorldTail*>(state + (size_t)w * stride + grid_pad);
  tail->seed = seeds[w];
  tail->episode = 0;
  tail->orders_step = 0;   // (the orders finish() left were drawn under the old seed)
}

// Sums the per-world event counters (WorldTail::ctr, reward_fx) over the shard.
__global__ void k_sum_counters(const uint8_t* state, int stride, int grid_pad, int n,
                               unsigned long long* out) {
  unsigned long long acc[MP_CTR_COUNT];
  for (int k = 0; k < MP_CTR_COUNT; ++k) acc[k] = 0;
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < n; w += gridDim.x * blockDim.x) {
    const WorldTail* tail = reinterpret_cast<const WorldTail*>(state + (size_t)w * stride + grid_pad);
    for (int k = 0; k < MP_CTR_COUNT; ++k) acc[k] += tail->ctr[k];
    acc[MP_CTR_REWARD_SUM] += (unsigned long long)(long long)tail->reward_fx;  // signed
  }
  for (int k = 0; k < MP_CTR_COUNT; ++k) {
    for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off);
    if ((threadIdx.x & 63) == 0 && acc[k]) atomicAdd(&out[k], acc[k]);
  }
}

int find_name(const void* pack, const char* table_name, const char* want) {
  uint64_t n = 0;
  const char* names = table<char>(pack, table_name, &n);
  int idx = 0;
  for (uint64_t i = 0; i < n; ++idx) {
    if (strcmp(names + i, want) == 0) return idx;
    i += strlen(names + i) + 1;
  }
  return -1;
}

// Every table the engine dereferences: present, of the right type, long enough,
// its indices in range — a truncated or stale pack is MP_ERR_PACK, never a wild
// pointer.  Host-only: runs before a device is touched.
int check_pack_tables(const void* hp, const int32_t* hdr) {
  {
    const int H = hdr[MPK_HDR_H], W = hdr[MPK_HDR_W], L = hdr[MPK_HDR_L];
    const int HW = H * W, NS = hdr[MPK_HDR_NSTATES], NSP = hdr[MPK_HDR_NSPRITES], PP = hdr[MPK_HDR_P];
    const int nobj = hdr[MPK_HDR_NOBJ], nhits = hdr[MPK_HDR_NHITS], nact = hdr[MPK_HDR_NACT];
    const int vl = hdr[MPK_HDR_VL], vr = hdr[MPK_HDR_VR], vf = hdr[MPK_HDR_VF], vb = hdr[MPK_HDR_VB];
    const int topology = hdr[MPK_HDR_TOPOLOGY], avatar_layer = hdr[MPK_HDR_AVATAR_LAYER];
    const int nf = hdr[MPK_HDR_NFIELDS];
    const int32_t* as = nf >= 1 && nf <= 4 ? table_n<int32_t>(hp, "action_spec", 3 * (uint64_t)nf)
                                           : nullptr;
    if (!as || !table<char>(hp, "action_names"))
      return fail(MP_ERR_PACK, "mp_create: the pack has no action_spec / action_names "
                               "(re-lower it with tools/make_packs.py)");
    for (int a = 0; a < nf; ++a)
      // (field 3 travels in six unsigned bits of the packed row: mp_step_fields)
      if (as[3 * a] < (a == 3 ? 0 : -128) || as[3 * a] > as[3 * a + 2] ||
          as[3 * a + 2] > as[3 * a + 1] || as[3 * a + 1] > (a == 3 ? 63 : 127))
        return fail(MP_ERR_PACK, "mp_create: action_spec field %d out of range", a);
    if (NS < 1 || NSP < 2 || nhits < 0 || nobj < 1 || hdr[MPK_HDR_MAXFRAMES] < 1 ||
        avatar_layer < 0 || avatar_layer >= L || vl < 0 || vr < 0 || vf < 0 ||
        vb < 0 || (vl + vr + 1) > 64 || (vf + vb + 1) > 64 ||
        (topology != 0 && topology != 1))
      return fail(MP_ERR_PACK, "mp_create: header fields out of range");
    {
      const int reach = std::max(std::max(vl, vr), std::max(vf, vb));
      if (topology == 1 && (reach > H || reach > W))   // the renderer wraps a coordinate once
        return fail(MP_ERR_PACK, "mp_create: a TORUS map smaller than the view's reach");
    }
    const uint8_t* ig = table_n<uint8_t>(hp, "init_grid", (uint64_t)L * HW);
    const int32_t* sl = table_n<int32_t>(hp, "state_layer", NS);
    const int32_t* ss = table_n<int32_t>(hp, "state_sprite", NS);
    const int32_t* so = table_n<int32_t>(hp, "state_orient", NS);
    const uint32_t* sg = table_n<uint32_t>(hp, "state_groups", NS);
    const uint32_t* hb = table_n<uint32_t>(hp, "state_hit_block", NS);
    const int32_t* al = table_n<int32_t>(hp, "avatar_alive_state", PP);
    const int32_t* wa = table_n<int32_t>(hp, "avatar_wait_state", PP);
    const int32_t* at = table_n<int32_t>(hp, "action_table", (uint64_t)nact * 4);
    const int32_t* hs = table_n<int32_t>(hp, "hit_state", nhits);
    const int32_t* hd = table_n<int32_t>(hp, "hit_state_dir", (uint64_t)nhits * 4);
    const uint8_t* rgba = table_n<uint8_t>(hp, "sprite_rgba", (uint64_t)NSP * 4 * 256);
    const int32_t* sf = table_n<int32_t>(hp, "sprite_flags", NSP);
    const int32_t* vm = table_n<int32_t>(hp, "view_sprite_map", (uint64_t)(PP + 1) * NSP);
    const int32_t* ob = table_n<int32_t>(hp, "objects", (uint64_t)nobj * 4);
    uint64_t nsc = 0;
    const int32_t* sc = table<int32_t>(hp, "spawn_cells", &nsc);
    if (!ig || !sl || !ss || !so || !sg || !hb || !al || !wa || !at || !hs || !hd || !rgba ||
        !sf || !vm || !ob || !sc || !table<char>(hp, "state_names") || !table<char>(hp, "hit_names"))
      return fail(MP_ERR_PACK, "mp_create: a table of the pack is missing, mistyped or too short "
                               "(re-lower it with tools/make_packs.py)");
    bool ok = in_range(sl, NS, -1, L) && in_range(ss, NS, -1, NSP) && in_range(so, NS, 0, 4) &&
              in_range(al, PP, 1, NS) && in_range(wa, PP, 1, NS) &&
              in_range(at, (uint64_t)nact * 4, -4, 5) && in_range(hs, nhits, 1, NS) &&
              in_range(hd, (uint64_t)nhits * 4, 1, NS) &&
              in_range(vm, (uint64_t)(PP + 1) * NSP, 0, NSP) && in_range(sc, nsc, 0, HW);
    for (uint64_t i = 0; ok && i < (uint64_t)L * HW; ++i) ok = ig[i] < NS;
    for (int i = 0; ok && i < nobj; ++i)
      ok = ob[4 * i + 1] >= 0 && ob[4 * i + 1] < W && ob[4 * i + 2] >= 0 && ob[4 * i + 2] < H &&
           ob[4 * i + 3] >= 1 && ob[4 * i + 3] < NS;
    if (!ok) return fail(MP_ERR_PACK, "mp_create: a table of the pack holds an index out of range");

    // the level's own tables: presence, type, length; cell lists inside the map
    struct Need { const char* name; uint32_t dtype; uint64_t min_count; };
    struct Cells { const char* name; uint64_t max_count; };
    std::vector<Need> need = {{"init_spawn_cells", MPK_I32, 1}, {"init_spawn_ptr", MPK_I32, 2},
                              {"avatar_init_group", MPK_I32, (uint64_t)PP},
                              {"init_spawn_mask", MPK_U32, 1}};
    std::vector<Cells> cells;
    const uint64_t P2 = (uint64_t)PP;
    switch (hdr[MPK_HDR_SUBSTRATE]) {
      case MPK_SUBSTRATE_CLEAN_UP:
        need.insert(need.end(), {{"cu_states", MPK_I32, 8}, {"cu_i32", MPK_I32, 7},
                                 {"cu_f64", MPK_F64, 6}, {"thr_misc", MPK_U64, 2},
                                 {"apple_thr", MPK_U64, 1}, {"zapper_i32", MPK_I32, 5},
                                 {"zapper_f64", MPK_F64, 2}});
        cells = {{"apple_cells", 256}, {"dirt_cells", 256}, {"water_cells", 256}};
        break;
      case MPK_SUBSTRATE_COMMONS_HARVEST:
        need.insert(need.end(), {{"ch_states", MPK_I32, 5}, {"ch_i32", MPK_I32, 4},
                                 {"ch_f64", MPK_F64, 1}, {"ch_thr", MPK_U64, 2},
                                 {"disc_offsets", MPK_I32, 2}, {"zapper_i32", MPK_I32, 5},
                                 {"zapper_f64", MPK_F64, 2}});
        cells = {{"apple_cells", 256}};
        break;
      case MPK_SUBSTRATE_TERRITORY:
        need.insert(need.end(), {{"tr_states", MPK_I32, 10 + 2 * P2}, {"tr_i32", MPK_I32, 16},
                                 {"tr_f64", MPK_F64, 8}, {"tr_thr", MPK_U64, 3},
                                 {"tr_hits", MPK_I32, 1 + 2 * P2}, {"zapper_i32", MPK_I32, 5},
                                 {"zapper_f64", MPK_F64, 2}});
        cells = {{"resource_cells", 256}};
        break;
      case MPK_SUBSTRATE_COINS:
        need.insert(need.end(), {{"co_states", MPK_I32, 3}, {"co_i32", MPK_I32, 4},
                                 {"co_f64", MPK_F64, 8}, {"co_thr", MPK_U64, 2}});
        cells = {{"coin_cells", 512}};
        break;
      case MPK_SUBSTRATE_COOP_MINING:
        need.insert(need.end(), {{"cm_states", MPK_I32, 5}, {"cm_i32", MPK_I32, 10},
                                 {"cm_f64", MPK_F64, 4 * P2}, {"cm_thr", MPK_U64, 3}});
        cells = {{"ore_cells", 640}};
        break;
      case MPK_SUBSTRATE_COLLABORATIVE_COOKING:
        need.insert(need.end(), {{"cc_inv_states", MPK_I32, 4}, {"cc_i32", MPK_I32, 3},
                                 {"cc_f64", MPK_F64, 1}, {"cc_pot_states", MPK_I32, 5},
                                 {"cc_bar_states", MPK_I32, 11}, {"cc_hits", MPK_I32, P2},
                                 {"cc_state_kind", MPK_U8, (uint64_t)hdr[MPK_HDR_NSTATES]}});
        cells = {{"cc_container_cells", 128}, {"cc_pot_cells", 64}, {"cc_receiver_cells", 64}};
        break;
      case MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS:
        need.insert(need.end(), {{"em_states", MPK_I32, 8}, {"em_i32", MPK_I32, 30},
                                 {"em_f64", MPK_F64, 8}, {"em_thr", MPK_U64, 21},
                                 {"zapper_i32", MPK_I32, 5}, {"zapper_f64", MPK_F64, 2}});
        cells = {{"mushroom_cells", 256}};
        break;
      case MPK_SUBSTRATE_GIFT_REFINEMENTS:
        need.insert(need.end(), {{"gr_states", MPK_I32, 2}, {"gr_i32", MPK_I32, 10},
                                 {"gr_f64", MPK_F64, 2 * P2 + 3}, {"gr_thr", MPK_U64, 2}});
        cells = {{"token_cells", 640}};
        break;
      case MPK_SUBSTRATE_THE_MATRIX: {
        // the table lengths follow from R (resource classes) and the number of
        // colour intervals, both in mx_i32
        const int32_t* mi = nullptr;
        if (!mpk_require(hp, "mx_i32", MPK_I32, 22, nullptr) ||
            !(mi = table<int32_t>(hp, "mx_i32")) || mi[0] < 1 || mi[0] > 3 || mi[19] < 1 ||
            mi[19] > 5 || mi[16] <= 0 || mi[18] < 1 || mi[18] > 3 || mi[21] < 0 || mi[21] >= nhits)
          return fail(MP_ERR_PACK, "mp_create: table 'mx_i32' is missing or holds constants out of range");
        const uint64_t R2 = (uint64_t)mi[0], NI = (uint64_t)mi[19];
        need.insert(need.end(), {{"mx_states", MPK_I32, 8 + 2 * R2},
                                 {"mx_f64", MPK_F64, 5 + 2 * R2 * R2 + 2 * NI},
                                 {"mx_thr", MPK_U64, 2},
                                 {"mx_player_i32", MPK_I32, 4 * P2},
                                 {"mx_player_f64", MPK_F64, 4 * P2},
                                 {"resource_class", MPK_I32, 1}});
        cells = {{"resource_cells", 128}};
        uint64_t ncl = 0, ncell = 0, nst = 0;
        const int32_t* cls = table<int32_t>(hp, "resource_class", &ncl);
        (void)table<int32_t>(hp, "resource_cells", &ncell);
        const int32_t* st = table<int32_t>(hp, "mx_states", &nst);
        if (!cls || ncl != ncell || !in_range(cls, ncl, 1, (int)R2 + 1))
          return fail(MP_ERR_PACK, "mp_create: table 'resource_class' does not match 'resource_cells'");
        if (!st || nst != 8 + 2 * R2 || !in_range(st, nst, 1, NS))
          return fail(MP_ERR_PACK, "mp_create: table 'mx_states' holds a state out of range");
        break;
      }
    }
    for (const Need& nd : need)
      if (!mpk_require(hp, nd.name, nd.dtype, nd.min_count, nullptr))
        return fail(MP_ERR_PACK, "mp_create: table '%s' is missing, mistyped or too short", nd.name);
    for (const Cells& cl : cells) {
      uint64_t cnt = 0;
      const int32_t* v = table<int32_t>(hp, cl.name, &cnt);
      if (!v || cnt > cl.max_count || !in_range(v, cnt, 0, HW))
        return fail(MP_ERR_PACK, "mp_create: table '%s' is missing, too long or leaves the map", cl.name);
    }
  }
  return MP_OK;
}

// Waits for the engine's stream and reports a frame kernel that gave up on its
// pipeline (frame.hip: report_stall) — an engine bug, surfaced instead of hung on.
int sync_and_check(MpEngine* e, const char* who) {
  HIP_TRY(hipStreamSynchronize(e->stream));
  const volatile uint32_t* f = e->h_fault;
  if (f[0] != 0)
    return fail(MP_ERR_HIP,
                "%s: the frame kernel's pipeline stalled (site %u, workgroup %u, wave %u, batch %u, "
                "seen %u, wanted %u); its outputs are incomplete",
                who, f[0], f[1], f[2], f[3], f[4], f[5]);
  if (f[8] != 0) {
    const uint32_t world = f[8] - 1;
    e->h_fault[8] = 0;   // reported once; the engine stays usable
    return fail(MP_ERR_HIP,
                "%s: world %u paid an interaction reward outside every resultIndicatorColorInterval "
                "(the reference asserts there, the_matrix/components.lua:282-290); the indicator "
                "shows the first colour", who, world);
  }
  return MP_OK;
}

// Draw-only launch: the views of the records as they are.
void draw(MpEngine* e, uint8_t* rgb, uint8_t* wrgb) {
  stepk::StepArgs args = {};
  args.state = e->d_state; args.num_worlds = e->N;
  FramePlan p = e->plan[0][rgb && wrgb ? 2 : wrgb ? 1 : 0];
  p.parity = e->frame_launches++ & 1;
  launch_frame(e->t, nullptr, args, rgb, wrgb, p, e->stream);
}

int submit(MpEngine* e, int mode, const int32_t* actions, const uint8_t* mask) {
  stepk::StepArgs args;
  args.state = e->d_state; args.actions = actions; args.reset_mask = mask;
  args.mode = mode; args.auto_reset = e->auto_reset; args.num_worlds = e->N;
  args.next_orders = e->next_orders;
  // the rollout ring: this submission's slot (a pointer store per ring-bound kind)
  const bool ringing = e->ring_slots > 0 && !e->ring_hold;
  const int slot = e->ring_slots > 0 ? (int)(e->ring_cursor % (uint64_t)e->ring_slots) : 0;
  if (ringing) { e->point_ring(slot); ++e->ring_cursor; }
  args.out = e->outputs();
  // One persistent launch steps the worlds and renders the bound views — one or
  // both — from the records while they are in LDS (frame.hip).
  uint8_t* rgb = (uint8_t*)e->bound[MP_OBS_RGB];
  uint8_t* wrgb = (uint8_t*)e->bound[MP_OBS_WORLD_RGB];
  const int views = rgb && wrgb ? 2 : wrgb ? 1 : 0;
  if ((!rgb && !wrgb) || !e->fuse(rgb == nullptr)) {
    launch_step(e->t, e->sub, args, e->stream);
    if (rgb) draw(e, rgb, nullptr);
    if (wrgb) draw(e, nullptr, wrgb);
  } else {
    // (the plan follows the buffer: a ring remembers one per slot, mp_tune)
    FramePlan p = ringing && e->ring_plan[views].size() == (size_t)e->ring_slots
                      ? e->ring_plan[views][(size_t)slot] : e->plan[1][views];
    p.parity = e->frame_launches++ & 1;
    launch_frame(e->t, &e->sub, args, rgb, wrgb, p, e->stream);
  }
  // "N.LAYER", when bound: one more (small) launch on the stepped records
  if (e->bound[MP_OBS_LAYER])
    launch_layer_view(e->t, e->d_state, (int32_t*)e->bound[MP_OBS_LAYER], e->N, e->stream);
  HIP_TRY(hipGetLastError());
  return MP_OK;
}

void retired_va(int64_t* bytes, int64_t* limit);   // (mapped views, below)
bool in_mapped_view(const void* p);

// A bound buffer is written by every launch from then on: a pointer the device cannot
// write (a host array, a stale tensor) would fault the GPU in the middle of a step — it is
// refused here instead.  Memory this library mapped itself is known by range (the runtime's
// pointer query does not know virtual-memory mappings).
int check_device_pointer(MpEngine* e, const void* ptr, const char* who) {
  if (in_mapped_view(ptr)) return MP_OK;
  hipPointerAttribute_t attr = {};
  const hipError_t rc = hipPointerGetAttributes(&attr, ptr);
  if (rc != hipSuccess) {
    (void)hipGetLastError();
    // The pointer query does not know virtual-memory mappings: a range ANOTHER library mapped
    // (torch's expandable segments, somebody's own hipMemMap) fails it although the device
    // writes it fine.  Such a range does answer hipMemGetAddressRange; only a pointer neither
    // query knows is refused.
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr) == hipSuccess && base && size) return MP_OK;
    (void)hipGetLastError();
    return fail(MP_ERR_INVALID, "%s: %p is not memory the device can write (%s); bind a device buffer",
                who, ptr, hipGetErrorString(rc));
  }
  if (attr.type == hipMemoryTypeUnregistered)
    return fail(MP_ERR_INVALID, "%s: %p is plain host memory; bind a device buffer", who, ptr);
  if ((attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeArray) && attr.device != e->device)
    return fail(MP_ERR_INVALID, "%s: %p lives on device %d, the engine on device %d", who, ptr,
                attr.device, e->device);
  return MP_OK;
}

// mp_bind_output on a kind that was bound as a ring: the kind leaves the ring
void drop_ring_kind(MpEngine* e, int kind) {
  if (!e->ring[kind].base) return;
  e->ring[kind] = MpEngine::RingKind();
  for (auto& v : e->ring_plan) v.clear();
  bool any = false;
  for (int k = 0; k < MP_OBS_KINDS; ++k) any = any || e->ring[k].base;
  if (!any) { e->ring_slots = 0; e->ring_cursor = 0; }
}


}  // namespace

extern "C" {

int mp_abi_version(void) { return MP_ABI_VERSION; }

const char* mp_last_error(void) { return g_error.c_str(); }

uint64_t mp_obs_bytes(const MpEngine* e, MpObsKind kind) {
  if (!e) return 0;
  const uint64_t N = (uint64_t)e->N, P = (uint64_t)e->t.P, S = (uint64_t)e->t.sprite_size;
  switch (kind) {
    case MP_OBS_RGB:
      return N * P * (e->t.vf + e->t.vb + 1) * S * (e->t.vl + e->t.vr + 1) * S * 3;
    case MP_OBS_WORLD_RGB: return N * e->t.H * S * e->t.W * S * 3;
    case MP_OBS_REWARD: case MP_OBS_READY_TO_SHOOT: case MP_OBS_AUX0: return N * P * 8;
    case MP_OBS_STEP_TYPE: return N * 4;
    case MP_OBS_DISCOUNT: case MP_OBS_COLLECTIVE_REWARD: return N * 8;
    case MP_OBS_POSITION: return N * P * 8;
    case MP_OBS_ORIENTATION: return N * P * 4;
    case MP_OBS_EVENTS: return N * MP_EVENT_ROWS * 16;
    case MP_OBS_AUX1: case MP_OBS_AUX2: case MP_OBS_AUX3: case MP_OBS_AUX4:
      return e->substrate == MPK_SUBSTRATE_CLEAN_UP ? N * P * 8 : 0;
    case MP_OBS_ZAP_MATRIX:
      return (e->substrate == MPK_SUBSTRATE_CLEAN_UP ||
              e->substrate == MPK_SUBSTRATE_COMMONS_HARVEST) ? N * P * P * 8 : 0;
    case MP_OBS_LAYER:
      return N * P * (e->t.vf + e->t.vb + 1) * (e->t.vl + e->t.vr + 1) * e->t.L * 4;
    case MP_OBS_INVENTORY:
      return N * P * e->inventory_types() * 8;
    case MP_OBS_INTERACTION_INVENTORIES:
      return e->substrate == MPK_SUBSTRATE_THE_MATRIX ? N * P * 2 * e->sub.mx.R * 8 : 0;
    case MP_OBS_MATRIX_CUMULANTS:
      return e->substrate == MPK_SUBSTRATE_THE_MATRIX ? N * P * (1 + 3 * e->sub.mx.R) * 8 : 0;
    case MP_OBS_INTERACTION_REWARDS:
      return e->substrate == MPK_SUBSTRATE_THE_MATRIX ? N * P * 2 * 8 : 0;
    default: return 0;
  }
}

static int create_impl(MpEngine* e, const void* pack, uint64_t pack_len,
                       const MpConfig* cfg);

int mp_create(const void* pack, uint64_t pack_len, const MpConfig* cfg,
              MpEngine** out) {
  if (!out) return fail(MP_ERR_INVALID, "mp_create: out is NULL");
  *out = nullptr;
  if (!cfg || cfg->struct_size != sizeof(MpConfig))
    return fail(MP_ERR_INVALID, "mp_create: bad MpConfig (struct_size)");
  if (cfg->dev && cfg->dev->struct_size != sizeof(MpDevOptions))
    return fail(MP_ERR_INVALID, "mp_create: bad MpDevOptions (struct_size)");
  if (cfg->num_worlds <= 0)
    return fail(MP_ERR_INVALID, "mp_create: num_worlds must be positive");
  if (mpk_validate(pack, pack_len) != 0)
    return fail(MP_ERR_PACK, "mp_create: not a valid MPK1 pack");
  const int32_t* hdr = table_n<int32_t>(pack, "hdr", MPK_HDR_LEN);
  if (!hdr || hdr[MPK_HDR_VERSION] != 1)
    return fail(MP_ERR_PACK, "mp_create: unsupported pack version");
  if (hdr[MPK_HDR_SUBSTRATE] != MPK_SUBSTRATE_CLEAN_UP &&
      hdr[MPK_HDR_SUBSTRATE] != MPK_SUBSTRATE_COMMONS_HARVEST &&
      hdr[MPK_HDR_SUBSTRATE] != MPK_SUBSTRATE_TERRITORY &&
      hdr[MPK_HDR_SUBSTRATE] != MPK_SUBSTRATE_COINS &&
      hdr[MPK_HDR_SUBSTRATE] != MPK_SUBSTRATE_THE_MATRIX &&
      hdr[MPK_HDR_SUBSTRATE] != MPK_SUBSTRATE_COOP_MINING &&
      hdr[MPK_HDR_SUBSTRATE] != MPK_SUBSTRATE_GIFT_REFINEMENTS &&
      hdr[MPK_HDR_SUBSTRATE] != MPK_SUBSTRATE_COLLABORATIVE_COOKING &&
      hdr[MPK_HDR_SUBSTRATE] != MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS)
    return fail(MP_ERR_PACK, "mp_create: substrate %d is not supported by this build",
                hdr[MPK_HDR_SUBSTRATE]);
  if (hdr[MPK_HDR_P] > MP_MAX_PLAYERS || hdr[MPK_HDR_P] < 1 || hdr[MPK_HDR_SPRITE] != 8 ||
      hdr[MPK_HDR_NSTATES] > 255 || hdr[MPK_HDR_NSPRITES] > 255 || hdr[MPK_HDR_NHITS] > 24 ||
      hdr[MPK_HDR_H] < 1 || hdr[MPK_HDR_W] < 1 || hdr[MPK_HDR_H] * hdr[MPK_HDR_W] > 4096 ||
      hdr[MPK_HDR_L] < 1 || hdr[MPK_HDR_NACT] < 1)
    return fail(MP_ERR_PACK, "mp_create: pack exceeds engine limits");
  if (cfg->num_players < 0 || cfg->num_players > hdr[MPK_HDR_P])
    return fail(MP_ERR_INVALID, "mp_create: num_players %d, the pack holds %d avatars",
                cfg->num_players, hdr[MPK_HDR_P]);

  if (int rc = check_pack_tables(pack, hdr)) return rc;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(MP_ERR_NO_DEVICE,
                "mp_create: no HIP device; the engine has no CPU path");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(MP_ERR_INVALID, "mp_create: device %d out of range (%d devices)",
                cfg->device, ndev);
  HIP_TRY(hipSetDevice(cfg->device));

  MpEngine* e = new MpEngine();
  const int rc = create_impl(e, pack, pack_len, cfg);
  if (rc != MP_OK) {
    mp_destroy(e);
    return rc;
  }
  *out = e;
  return MP_OK;
}

static int create_impl(MpEngine* e, const void* pack, uint64_t pack_len,
                       const MpConfig* cfg) {
  const int32_t* hdr = nullptr;
  const MpDevOptions* dev = cfg->dev;   // tests / tools only (include/mp_engine.h)
  e->has_dev = dev != nullptr;
  e->next_orders = !(dev && dev->no_next_orders);
  e->device = cfg->device;
  e->N = cfg->num_worlds;
  e->auto_reset = cfg->auto_reset;
  e->stream = (hipStream_t)cfg->stream;
  if (cfg->unfused < 0 || cfg->unfused > 2)
    return fail(MP_ERR_INVALID, "mp_create: MpConfig.unfused must be 0, 1 or 2 (got %d)", cfg->unfused);
  e->unfused = cfg->unfused;   // 0 is resolved once the pack is read
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess ||
        cus <= 0)
      return fail(MP_ERR_NO_DEVICE, "mp_create: device %d does not report its compute units", cfg->device);
    e->num_cus = cus;
  }
  e->pack.assign((const uint8_t*)pack, (const uint8_t*)pack + pack_len);
  const void* hp = e->pack.data();
  hdr = table<int32_t>(hp, "hdr");
  e->substrate = hdr[MPK_HDR_SUBSTRATE];
  e->sub.substrate = e->substrate;

  DevTables& t = e->t;
  t.H = hdr[MPK_HDR_H]; t.W = hdr[MPK_HDR_W]; t.L = hdr[MPK_HDR_L];
  t.P_pack = hdr[MPK_HDR_P];
  t.P = cfg->num_players > 0 ? cfg->num_players
        : hdr[MPK_HDR_DEFAULT_P] > 0 && hdr[MPK_HDR_DEFAULT_P] <= t.P_pack ? hdr[MPK_HDR_DEFAULT_P]
                                                                             : t.P_pack;
  t.nstates = hdr[MPK_HDR_NSTATES];
  t.nsprites = hdr[MPK_HDR_NSPRITES]; t.topology = hdr[MPK_HDR_TOPOLOGY];
  t.max_frames = hdr[MPK_HDR_MAXFRAMES]; t.nact = hdr[MPK_HDR_NACT];
  if (cfg->roles) {
    // Per-player constants by role (bach_or_stravinsky: create_avatar_objects(roles),
    // bach_or_stravinsky_in_the_matrix__repeated.py:473-497): the pack holds, per
    // (role, player), the avatar's sprite and its row of mx_player_*; this engine's
    // copy of the pack becomes the one lowered for the requested assignment
    // (meltingpot_amd/lower.py: add_role_tables / apply_roles), before anything
    // is derived from it.
    uint64_t n_names = 0, n_rgba = 0, n_pi = 0, n_pf = 0;
    const char* names = table<char>(hp, "role_names", &n_names);
    const int32_t* sprite = table_n<int32_t>(hp, "role_sprite", t.P_pack);
    const uint8_t* rgba = table<uint8_t>(hp, "role_rgba", &n_rgba);
    const int32_t* rpi = table<int32_t>(hp, "role_player_i32", &n_pi);
    const double* rpf = table<double>(hp, "role_player_f64", &n_pf);
    int n_roles = 0;
    for (uint64_t i = 0; names && i < n_names; ++i) n_roles += names[i] == 0;
    const size_t block = (size_t)4 * hdr[MPK_HDR_SPRITE] * hdr[MPK_HDR_SPRITE] * 4;
    const size_t PP = (size_t)t.P_pack;
    uint64_t n_srgba = 0, n_mpi = 0, n_mpf = 0;
    uint8_t* srgba = const_cast<uint8_t*>(table<uint8_t>(hp, "sprite_rgba", &n_srgba));
    int32_t* mpi = const_cast<int32_t*>(table<int32_t>(hp, "mx_player_i32", &n_mpi));
    double* mpf = const_cast<double*>(table<double>(hp, "mx_player_f64", &n_mpf));
    if (n_roles < 1 || !sprite || !rgba || !rpi || !rpf || !srgba || !mpi || !mpf ||
        n_rgba != n_roles * PP * block || n_pi != n_roles * PP * 4 || n_pf != n_roles * PP * 4 ||
        n_mpi < PP * 4 || n_mpf < PP * 4 || !in_range(sprite, PP, 0, t.nsprites) ||
        n_srgba < (size_t)t.nsprites * block)
      return fail(MP_ERR_INVALID, "mp_create: MpConfig.roles given, but this substrate's pack holds "
                                  "no per-role tables (its config has one valid role)");
    for (int p = 0; p < t.P; ++p) {
      const int r = cfg->roles[p];
      if (r < 0 || r >= n_roles)
        return fail(MP_ERR_INVALID, "mp_create: role %d of player %d is outside [0, %d)", r, p + 1,
                    n_roles);
      memcpy(srgba + (size_t)sprite[p] * block, rgba + ((size_t)r * PP + p) * block, block);
      memcpy(mpi + 4 * p, rpi + ((size_t)r * PP + p) * 4, 4 * sizeof(int32_t));
      memcpy(mpf + 4 * p, rpf + ((size_t)r * PP + p) * 4, 4 * sizeof(double));
    }
  }
  {
    // raw action fields (mp_step_fields): actionSpec (min, max, default) per field
    const int32_t* spec = table_n<int32_t>(hp, "action_spec", 3 * (uint64_t)hdr[MPK_HDR_NFIELDS]);
    t.nfields = hdr[MPK_HDR_NFIELDS];
    t.field_lo = t.field_hi = 0;
    for (int a = 0; a < t.nfields; ++a) {
      t.field_lo |= ((uint32_t)spec[3 * a] & 255u) << (8 * a);
      t.field_hi |= ((uint32_t)spec[3 * a + 1] & 255u) << (8 * a);
    }
  }
  t.avatar_layer = hdr[MPK_HDR_AVATAR_LAYER]; t.sprite_size = hdr[MPK_HDR_SPRITE];
  t.vl = hdr[MPK_HDR_VL]; t.vr = hdr[MPK_HDR_VR];
  t.vf = hdr[MPK_HDR_VF]; t.vb = hdr[MPK_HDR_VB];
  // territory keeps three per-cell resource planes behind the render planes, the
  // matrix levels two and a block of per-player variables (step_matrix.h)
  t.grid_planes = t.L + (hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_TERRITORY ? 3 : 0) +
                  (hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_THE_MATRIX ? 2 : 0) +
                  (hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COOP_MINING ? 2 : 0) +
                  (hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COLLABORATIVE_COOKING ? 1 : 0) +
                  (hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS ? 1 : 0);
  t.grid_bytes = t.grid_planes * t.H * t.W;
  if (hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_THE_MATRIX) {
    e->mx.player_block = (t.grid_bytes + 15) & ~15;
    t.grid_bytes = e->mx.player_block + MP_MAX_PLAYERS * (int)sizeof(stepk::MxPlayer);
  }
  t.grid_pad = (t.grid_bytes + 15) & ~15;
  t.world_stride = ((t.grid_pad + (int)sizeof(WorldTail) + 63) & ~63) +
                   64 * (dev && dev->record_pad > 0 ? dev->record_pad : 0);
  e->nhits = hdr[MPK_HDR_NHITS];
#define DEV_ALLOC(ptr, bytes) HIP_TRY(hipMalloc((void**)&(ptr), (bytes)))
  DEV_ALLOC(e->d_pack, pack_len);
  HIP_TRY(hipMemcpy(e->d_pack, hp, pack_len, hipMemcpyHostToDevice));

  uint64_t n = 0;
  t.init_grid = e->dev<uint8_t>(table<uint8_t>(hp, "init_grid"));
  t.state_layer = e->dev<int32_t>(table<int32_t>(hp, "state_layer"));
  t.state_sprite = e->dev<int32_t>(table<int32_t>(hp, "state_sprite"));
  t.alive_state = e->dev<int32_t>(table<int32_t>(hp, "avatar_alive_state"));
  t.wait_state = e->dev<int32_t>(table<int32_t>(hp, "avatar_wait_state"));
  t.action_table = e->dev<int32_t>(table<int32_t>(hp, "action_table"));
  const int32_t* spawn = table<int32_t>(hp, "spawn_cells", &n);
  t.spawn_cells = e->dev<int32_t>(spawn);
  t.n_spawn = (int)n;
  t.hit_state = e->dev<int32_t>(table<int32_t>(hp, "hit_state"));
  t.hit_state_dir = e->dev<int32_t>(table<int32_t>(hp, "hit_state_dir"));
  t.state_orient = e->dev<int32_t>(table<int32_t>(hp, "state_orient"));
  if (!table<int32_t>(hp, "hit_state_dir") || !table<int32_t>(hp, "state_orient"))
    return fail(MP_ERR_PACK, "mp_create: pack lacks hit_state_dir / state_orient (re-lower it)");
  t.sprite_rgba = e->dev<uint8_t>(table<uint8_t>(hp, "sprite_rgba"));
  t.view_sprite_map = e->dev<int32_t>(table<int32_t>(hp, "view_sprite_map"));
  t.state_groups = e->dev<uint32_t>(table<uint32_t>(hp, "state_groups"));
  {
    const int32_t* opt = table<int32_t>(hp, "optional_i32", &n);
    t.n_optional = opt ? (int)(n / 4) : 0;
    t.optional = opt ? e->dev<int32_t>(opt) : nullptr;
    const int32_t* cn = table<int32_t>(hp, "choice_n");
    t.choice_n = cn ? e->dev<int32_t>(cn) : nullptr;
    uint64_t ncn = 0;
    (void)table<int32_t>(hp, "choice_n", &ncn);
    if (t.n_optional > 0 && !cn)
      return fail(MP_ERR_PACK, "mp_create: optional objects without choice_n");
    if (opt && (n % 4) != 0) return fail(MP_ERR_PACK, "mp_create: optional_i32 is not [n][4]");
    for (uint64_t i = 0; i < ncn; ++i)
      if (cn[i] == 0 || cn[i] > 64 || cn[i] < -64 || ncn > 65535)
        return fail(MP_ERR_PACK, "mp_create: choice_n out of range");
    for (int i = 0; i < t.n_optional; ++i) {
      const int32_t* o4 = opt + 4 * i;   // cell, plane | initial state << 8, choice, outcome mask
      if (o4[0] < 0 || o4[0] >= t.H * t.W || o4[1] < 0 || (o4[1] & 255) >= t.L ||
          (o4[1] >> 8) < 1 || (o4[1] >> 8) >= t.nstates || o4[2] < 0 ||
          (uint64_t)(o4[2] & 0xffff) >= ncn || (o4[2] >> 16) > 32)
        return fail(MP_ERR_PACK, "mp_create: optional object %d out of range", i);
    }
  }
  if (t.n_spawn < t.P || t.n_spawn > 256)
    return fail(MP_ERR_PACK, "mp_create: %d spawn points for %d players", t.n_spawn, t.P);

  // derived tables: renderer sprite flags; state -> player
  {
    const int32_t* flags = table<int32_t>(hp, "sprite_flags");
    const int32_t* alive = table<int32_t>(hp, "avatar_alive_state");
    const int32_t* ssprite = table<int32_t>(hp, "state_sprite");
    const int32_t* slayer = table<int32_t>(hp, "state_layer");
    // [0,256) sprite flags, [256,512) state -> player, then u16 res_index[H*W]
    // (territory: cell -> index into resource_cells, 0xffff = none)
    std::vector<uint8_t> extra(512 + (size_t)t.H * t.W * 2, 0xff);
    memset(extra.data(), 0, 512);
    for (int s = 0; s < t.nsprites; ++s)
      extra[s] = (uint8_t)(((flags[s] & MPK_SPRITE_OPAQUE) ? 1 : 0) |
                           ((flags[s] & MPK_SPRITE_PARTIAL) ? 2 : 0));
    int8_t* sp = reinterpret_cast<int8_t*>(extra.data() + 256);
    for (int s = 0; s < 256; ++s) sp[s] = -1;
    for (int p = 0; p < t.P; ++p) sp[alive[p]] = (int8_t)p;
    uint64_t n_extra_alive = 0;
    const int32_t* extra_alive = table<int32_t>(hp, "avatar_extra_alive", &n_extra_alive);
    for (uint64_t i = 0; extra_alive && i + 1 < n_extra_alive; i += 2)
      if (extra_alive[i] > 0 && extra_alive[i] < 256 && extra_alive[i + 1] < t.P)
        sp[extra_alive[i]] = (int8_t)extra_alive[i + 1];
    // the renderer resolves non-avatar sprites through one table shared by all
    // viewers: only avatar sprites may be remapped per viewer (clean_up.py:630-631)
    {
      const int32_t* vmap = table<int32_t>(hp, "view_sprite_map");
      std::vector<uint8_t> is_avatar_sprite((size_t)t.nsprites, 0);
      for (int p = 0; p < t.P; ++p)
        if (ssprite[alive[p]] >= 0) is_avatar_sprite[(size_t)ssprite[alive[p]]] = 1;
      for (uint64_t i = 0; extra_alive && i + 1 < n_extra_alive; i += 2)
        if (extra_alive[i] > 0 && extra_alive[i] < t.nstates && ssprite[extra_alive[i]] >= 0)
          is_avatar_sprite[(size_t)ssprite[extra_alive[i]]] = 1;
      for (int v = 0; v < t.P; ++v)
        for (int s = 0; s < t.nsprites; ++s)
          if (!is_avatar_sprite[(size_t)s] && vmap[v * t.nsprites + s] != vmap[t.P_pack * t.nsprites + s])
            return fail(MP_ERR_PACK, "mp_create: viewer %d remaps non-avatar sprite %d", v, s);
    }
    // the renderer's draw list holds one opaque base + 8 overlays per cell
    int drawn_layers = 0;
    for (int l = 0; l < t.L; ++l) {
      bool any = false;
      for (int s = 1; s < t.nstates; ++s) any = any || (slayer[s] == l && ssprite[s] >= 0);
      drawn_layers += any;
    }
    // (collaborative_cooking has one interact layer per avatar, each showing a sprite on the
    // ONE cell its avatar faces: at most four of them meet on a cell)
    if (hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COLLABORATIVE_COOKING && t.P_pack > 4)
      drawn_layers -= t.P_pack - 4;
    if (drawn_layers > 9 || t.L > 12)
      return fail(MP_ERR_PACK, "mp_create: %d sprite-bearing layers that can meet on a cell (max 9), "
                               "%d layers (max 12)", drawn_layers, t.L);
    DEV_ALLOC(e->d_extra, extra.size());
    HIP_TRY(hipMemcpy(e->d_extra, extra.data(), extra.size(), hipMemcpyHostToDevice));
    t.sprite_flags8 = e->d_extra;
  }
  // the step kernels' LDS tables (step_common.h): per state the BeamBlocker bits
  // and the avatar it is the live state of; the respawn group's cells
  {
    const uint32_t* hb = table<uint32_t>(hp, "state_hit_block");
    const int32_t* alive = table<int32_t>(hp, "avatar_alive_state");
    if (!hb || !alive) return fail(MP_ERR_PACK, "mp_create: pack lacks state_hit_block");
    std::vector<uint8_t> blob((size_t)stepk::tables_bytes(t), 0);
    uint32_t* sinfo = reinterpret_cast<uint32_t*>(blob.data());
    for (int s2 = 0; s2 < t.nstates; ++s2) sinfo[s2] = hb[s2] & 0xffffffu;
    for (int p2 = 0; p2 < t.P; ++p2) {
      if (alive[p2] <= 0 || alive[p2] >= t.nstates)
        return fail(MP_ERR_PACK, "mp_create: avatar state out of range");
      sinfo[alive[p2]] |= (uint32_t)(p2 + 1) << 24;
    }
    {
      // more alive states of an avatar (coins: one per colour): (state, player) pairs
      uint64_t nx = 0;
      const int32_t* xa = table<int32_t>(hp, "avatar_extra_alive", &nx);
      for (uint64_t i = 0; xa && i + 1 < nx; i += 2) {
        if (xa[i] <= 0 || xa[i] >= t.nstates || xa[i + 1] < 0 || xa[i + 1] >= t.P_pack)
          return fail(MP_ERR_PACK, "mp_create: avatar_extra_alive out of range");
        if (xa[i + 1] < t.P) sinfo[xa[i]] |= (uint32_t)(xa[i + 1] + 1) << 24;
      }
    }
    uint16_t* sp16 = reinterpret_cast<uint16_t*>(blob.data() + stepk::kSinfoBytes);
    for (int i = 0; i < t.n_spawn; ++i) {
      if (spawn[i] < 0 || spawn[i] >= t.H * t.W)
        return fail(MP_ERR_PACK, "mp_create: spawn cell out of range");
      sp16[i] = (uint16_t)spawn[i];
    }
    const int32_t* at = table<int32_t>(hp, "action_table");
    int8_t* rows = reinterpret_cast<int8_t*>(blob.data() + stepk::kSinfoBytes +
                                             stepk::spawn_bytes(t.n_spawn));
    for (int i = 0; i < t.nact * 4; ++i) rows[i] = (int8_t)at[i];
    // (host memory: readable without a HIP call, i.e. while a kernel is stuck)
    // (64 fault words + the -DMP_FRAME_TIMELINE build's event log)
    HIP_TRY(hipHostMalloc((void**)&e->h_fault, kFaultWords * sizeof(uint32_t), hipHostMallocMapped));
    memset(e->h_fault, 0, kFaultWords * sizeof(uint32_t));
    HIP_TRY(hipHostGetDevicePointer((void**)&t.fault, e->h_fault, 0));
    DEV_ALLOC(e->d_claim, (2 + 2 * 1024) * sizeof(uint32_t));   // (+ the -DMP_FRAME_ENDS build's stamps)
    HIP_TRY(hipMemset(e->d_claim, 0, (2 + 2 * 1024) * sizeof(uint32_t)));
    t.claim = e->d_claim;
    DEV_ALLOC(e->d_stepblob, blob.size());
    HIP_TRY(hipMemcpy(e->d_stepblob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    t.step_blob = e->d_stepblob;
  }

  // ---- rules shared by every substrate with the stock avatar: Zapper kwargs,
  // beam footprint, spawn groups
  const int32_t* slayer = table<int32_t>(hp, "state_layer");
  const int32_t* hit_state = table<int32_t>(hp, "hit_state");
  // beam footprint in the order the reference walks it: the centre ray, then
  // for the left and the right side every lateral cell followed by the forward
  // ray that starts there
  auto make_shape = [&](int len, int rad, BeamShape* sh) -> int {
    int cnt = 0;
    auto add = [&](int lat, int fwd, uint32_t pred) {
      if (cnt < 16)
        sh->cell[cnt] = ((uint32_t)lat & 255u) | (((uint32_t)fwd & 255u) << 8) | ((pred & 0xffffu) << 16);
      return cnt++;
    };
    uint32_t pred = 0;
    for (int f = 1; f <= len; ++f) pred |= 1u << add(0, f, pred);
    for (int side = -1; side <= 1; side += 2) {
      uint32_t side_pred = 0;
      for (int i = 1; i <= rad; ++i) {
        side_pred |= 1u << add(side * i, 0, side_pred);
        uint32_t ray_pred = side_pred;
        for (int f = 1; f <= len - i; ++f) ray_pred |= 1u << add(side * i, f, ray_pred);
      }
    }
    sh->n = cnt;
    // (round 5: an integer division is ~45 instructions on this ISA, and the six of a
    // clean_up step — lane / n and 64 / n for either beam — were hoisted into the 894
    // instructions a feeder executes in front of its first world: profiles/r05_head.md)
    sh->per = cnt > 0 ? 64 / cnt : 0;
    sh->magic = cnt > 0 ? 65536u / (uint32_t)cnt + 1u : 0u;
    return cnt;
  };
  ZapRules zap{};
  // (coins avatars carry none, the matrix levels' GameInteractionZapper has its own tables)
  const bool has_zapper = e->substrate != MPK_SUBSTRATE_COINS &&
                          e->substrate != MPK_SUBSTRATE_THE_MATRIX &&
                          e->substrate != MPK_SUBSTRATE_COOP_MINING &&
                          e->substrate != MPK_SUBSTRATE_GIFT_REFINEMENTS &&
                          e->substrate != MPK_SUBSTRATE_COLLABORATIVE_COOKING;
  if (has_zapper) {
    const int32_t* zi = table<int32_t>(hp, "zapper_i32");
    const double* zf = table<double>(hp, "zapper_f64");
    zap.hit = find_name(hp, "hit_names", "zapHit");
    if (!zi || !zf || zap.hit < 0)
      return fail(MP_ERR_PACK, "mp_create: no Zapper tables in the pack");
    zap.cooldown = zi[0]; zap.length = zi[1]; zap.radius = zi[2];
    zap.respawn_frames = zi[3]; zap.remove_hit = zi[4];
    zap.penalty = zf[0]; zap.reward = zf[1];
    zap.s_hit = hit_state[zap.hit]; zap.layer = slayer[zap.s_hit];
    if (zap.cooldown > 255 || make_shape(zap.length, zap.radius, &zap.shape) > 16)
      return fail(MP_ERR_PACK, "mp_create: Zapper constants out of engine range");
  }
  {
    uint64_t ncells = 0;
    const int32_t* cells = table<int32_t>(hp, "init_spawn_cells", &ncells);
    const int32_t* ptr = table<int32_t>(hp, "init_spawn_ptr", &n);
    const int32_t* grp = table_n<int32_t>(hp, "avatar_init_group", t.P_pack);
    if (!cells || !ptr || !grp || n < 2 || n > 65)
      return fail(MP_ERR_PACK, "mp_create: no spawn group tables in the pack");
    t.n_init_groups = (int)n - 1;
    if (ptr[0] != 0 || (uint64_t)ptr[t.n_init_groups] != ncells ||
        !in_range(cells, ncells, 0, t.H * t.W) || !in_range(grp, t.P_pack, 0, t.n_init_groups))
      return fail(MP_ERR_PACK, "mp_create: spawn group tables inconsistent");
    uint64_t nm0 = 0;
    const uint32_t* masks0 = table<uint32_t>(hp, "init_spawn_mask", &nm0);
    // is any optional object a spawn point?  (then the reset filters the pools)
    t.optional_spawn = 0;
    {
      const uint32_t* sg = table<uint32_t>(hp, "state_groups");
      const int32_t* opt = table<int32_t>(hp, "optional_i32");
      for (int i = 0; i < t.n_optional && masks0 && sg; ++i)
        for (uint64_t g = 0; g < nm0; ++g)
          if (sg[opt[4 * i + 1] >> 8] & masks0[g]) t.optional_spawn = 1;
    }
    for (int g = 0; g < t.n_init_groups; ++g)
      if (ptr[g + 1] < ptr[g] ||
          ptr[g + 1] - ptr[g] > (t.optional_spawn ? 64
                                 // (step_mushroom.h: spawn_avatars_wide)
                                 : e->substrate == MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS ? 256 : 128))
        return fail(MP_ERR_PACK, "mp_create: too many cells in a spawn group (%d)",
                    ptr[g + 1] - ptr[g]);
    t.init_spawn_cells = e->dev<int32_t>(cells);
    t.init_spawn_ptr = e->dev<int32_t>(ptr);
    t.avatar_init_group = e->dev<int32_t>(grp);
    uint64_t nm = 0;
    const uint32_t* masks = table<uint32_t>(hp, "init_spawn_mask", &nm);
    if (!masks || (int)nm != t.n_init_groups)
      return fail(MP_ERR_PACK, "mp_create: pack lacks init_spawn_mask (re-lower it)");
    t.init_spawn_mask = e->dev<uint32_t>(masks);
    // every avatar that plays needs a point of its group (base_simulation.lua:
    // 396-445 "Insufficient spawn points!")
    for (int g = 0; g < t.n_init_groups; ++g) {
      int want = 0;
      for (int p2 = 0; p2 < t.P; ++p2) want += grp[p2] == g;
      if (!t.optional_spawn && want > ptr[g + 1] - ptr[g])
        return fail(MP_ERR_PACK, "mp_create: %d avatars for the %d points of spawn group %d",
                    want, ptr[g + 1] - ptr[g], g);
    }
    // (with 'choice' spawn points the respawn pool would have to be filtered by
    // presence: only substrates that never respawn are accepted)
    if (t.optional_spawn && (zap.remove_hit || e->substrate == MPK_SUBSTRATE_THE_MATRIX))
      return fail(MP_ERR_PACK, "mp_create: optional spawn points in a level that respawns");
  }
  auto only_beams_on = [&](int layer, int s_beam) {
    for (int s = 1; s < t.nstates; ++s)
      if (s != s_beam && slayer[s] == layer) return false;
    return true;
  };
  if (has_zapper && !only_beams_on(zap.layer, zap.s_hit))
    return fail(MP_ERR_PACK, "mp_create: a piece state lives on the zap beam layer");

  if (e->substrate == MPK_SUBSTRATE_THE_MATRIX) {
    MatrixTables& c = e->mx;
    const int32_t* st = table<int32_t>(hp, "mx_states", &n);
    const uint64_t nst = n;
    const int32_t* ci = table_n<int32_t>(hp, "mx_i32", 22);
    uint64_t nf = 0;
    const double* cf = table<double>(hp, "mx_f64", &nf);
    const uint64_t* thr = table_n<uint64_t>(hp, "mx_thr", 2);
    const int32_t* pi = table_n<int32_t>(hp, "mx_player_i32", 4 * (uint64_t)t.P_pack);
    const double* pf = table_n<double>(hp, "mx_player_f64", 4 * (uint64_t)t.P_pack);
    uint64_t ns = 0, ncl = 0;
    const int32_t* cells = table<int32_t>(hp, "resource_cells", &ns);
    const int32_t* cls = table<int32_t>(hp, "resource_class", &ncl);
    if (!st || !ci || !cf || !thr || !pi || !pf || !cells || !cls || ns != ncl || ns > 128)
      return fail(MP_ERR_PACK, "mp_create: the_matrix tables missing or out of engine range");
    const int R = ci[0];
    if (R < 1 || R > stepk::kMxMaxR || nst != (uint64_t)(8 + 2 * R) || ci[19] < 1 || ci[19] > 5 ||
        nf != (uint64_t)(5 + 2 * R * R + 2 * ci[19]) || !in_range(st, nst, 1, t.nstates) ||
        !in_range(cls, ncl, 1, R + 1) || !in_range(cells, ns, 0, t.H * t.W))
      return fail(MP_ERR_PACK, "mp_create: the_matrix tables inconsistent");
    c.R = R;
    c.n_site = (int)ns;
    c.site_cells = e->dev<int32_t>(cells); c.site_class = e->dev<int32_t>(cls);
    c.player_i32 = e->dev<int32_t>(pi); c.player_f64 = e->dev<double>(pf);
    c.cooldown = ci[1]; c.respawn_frames = ci[4]; c.freeze = ci[5]; c.end_on_first = ci[6];
    c.reset_winner = ci[7]; c.reset_loser = ci[8]; c.loser_dies = ci[9]; c.winner_dies = ci[10];
    c.zero_inventory = ci[11]; c.random_tie = ci[12]; c.disallow_unready = ci[13];
    c.has_ee = ci[14]; c.ee_min_frames = ci[15]; c.ee_interval = ci[16];
    c.regen_delay = ci[17]; c.initial_health = ci[18]; c.n_intervals = ci[19];
    c.spawn_all = ci[20]; c.hit = ci[21];
    c.reward_floor = cf[0]; c.reward_multiplier = cf[1]; c.reward_unready = cf[2];
    for (int i = 0; i < R * R; ++i) { c.row_matrix[i] = cf[5 + i]; c.col_matrix[i] = cf[5 + R * R + i]; }
    for (int i = 0; i < 2 * c.n_intervals; ++i) c.interval[i] = cf[5 + 2 * R * R + i];
    c.thr_regen = thr[0]; c.thr_ee = thr[1];
    {
      // TheMatrix:getColorInterval asserts that an interval holds the reward
      // (components.lua:282-290); the kernel cannot assert, so the pack must make
      // the assertion unreachable: a reward is rewardMultiplier x a convex
      // combination of matrix entries (or 0 with an empty inventory), and every
      // point of that range has to lie in one of the [lo, hi) intervals.
      double lo = 0.0, hi = 0.0;
      for (int i = 0; i < R * R; ++i)
        for (double v : {c.reward_multiplier * c.row_matrix[i], c.reward_multiplier * c.col_matrix[i]}) {
          lo = std::min(lo, v); hi = std::max(hi, v);
        }
      auto covered = [&](double x) {
        for (int k = 0; k < c.n_intervals; ++k)
          if (c.interval[2 * k] <= x && x < c.interval[2 * k + 1]) return true;
        return false;
      };
      // (the two extreme payoffs themselves need pure profiles on both sides;
      // the stock intervals end exactly there, half-open, and the reference would
      // assert if one were ever paid: the kernel reports that case through the
      // fault words instead — sync_and_check — and every other reward is checked
      // here: each stretch between neighbouring interval bounds inside (lo, hi))
      std::vector<double> cuts = {lo, hi};
      for (int k = 0; k < 2 * c.n_intervals; ++k)
        if (c.interval[k] > lo && c.interval[k] < hi) cuts.push_back(c.interval[k]);
      std::sort(cuts.begin(), cuts.end());
      bool ok = true;
      for (size_t i = 0; ok && i + 1 < cuts.size(); ++i) {
        if (cuts[i] == cuts[i + 1]) continue;
        ok = covered(0.5 * (cuts[i] + cuts[i + 1])) && (i == 0 || covered(cuts[i]));
      }
      if (!ok)
        return fail(MP_ERR_PACK, "mp_create: resultIndicatorColorIntervals do not cover the rewards "
                                 "(%g, %g) this matrix and rewardMultiplier can pay "
                                 "(the reference asserts, components.lua:282-290)", lo, hi);
    }
    if (c.hit < 0 || c.hit >= e->nhits || c.cooldown < 1 || c.cooldown > 255 || c.freeze < 0 ||
        c.freeze > 200 || c.initial_health < 1 || c.initial_health > 3 || c.ee_interval <= 0 ||
        c.respawn_frames < 0 || (c.regen_delay > 250 && c.thr_regen != 0) ||
        make_shape(ci[2], ci[3], &c.shape) > 16)
      return fail(MP_ERR_PACK, "mp_create: the_matrix constants out of engine range");
    if (c.regen_delay > 255) c.regen_delay = 255;   // (never reached: the rate is 0)
    c.s_beam = hit_state[c.hit]; c.beam_layer = slayer[c.s_beam];
    // marker states in indicator order: notReady, ready, colour 1..5 (mx_states:
    // wait, ready, notReady, colours); resource states per class: visible, wait
    const int mark_wait = st[0];
    const int by_ind[7] = {st[2], st[1], st[3], st[4], st[5], st[6], st[7]};
    c.s_mark_packed = 0;
    c.mark_layer = slayer[st[2]];
    for (int i = 0; i < 7; ++i) {
      c.s_mark_packed |= (uint64_t)by_ind[i] << (8 * i);
      // 'notReady' draws nothing but sits on the overlay layer like the others
      if (slayer[by_ind[i]] != c.mark_layer)
        return fail(MP_ERR_PACK, "mp_create: the_matrix marker states on different layers");
    }
    c.s_visible_packed = 0;
    c.res_layer = slayer[st[8]];
    for (int k = 0; k < R; ++k) {
      c.s_visible_packed |= (uint32_t)st[8 + 2 * k] << (8 * k);
      if (slayer[st[8 + 2 * k]] != c.res_layer || slayer[st[9 + 2 * k]] >= 0)
        return fail(MP_ERR_PACK, "mp_create: the_matrix resource states on unexpected layers");
    }
    if (slayer[mark_wait] >= 0 || c.mark_layer < 0 || c.res_layer < 0 || c.beam_layer < 0 ||
        c.mark_layer == t.avatar_layer || c.res_layer == t.avatar_layer ||
        !only_beams_on(c.beam_layer, c.s_beam))
      return fail(MP_ERR_PACK, "mp_create: the_matrix layers out of engine range");
    // the overlay layer holds markers only, the resource layer resources only
    for (int s2 = 1; s2 < t.nstates; ++s2) {
      bool is_mark = false, is_res = false;
      for (int i = 0; i < 7; ++i) is_mark = is_mark || s2 == by_ind[i];
      for (int k = 0; k < R; ++k) is_res = is_res || s2 == st[8 + 2 * k];
      if ((slayer[s2] == c.mark_layer && !is_mark) || (slayer[s2] == c.res_layer && !is_res))
        return fail(MP_ERR_PACK, "mp_create: the_matrix: a foreign state on the marker / resource layer");
    }
    c.plane_a = t.L; c.plane_b = t.L + 1;
    if (t.W > 255 || t.H > 255) return fail(MP_ERR_PACK, "mp_create: the_matrix map too large");
  }

  if (e->substrate == MPK_SUBSTRATE_COOP_MINING) {
    CoopTables& c = e->cm;
    const int32_t* st = table_n<int32_t>(hp, "cm_states", 5);
    const int32_t* ci = table_n<int32_t>(hp, "cm_i32", 10);
    const double* cf = table_n<double>(hp, "cm_f64", 4 * (uint64_t)t.P_pack);
    const uint64_t* thr = table_n<uint64_t>(hp, "cm_thr", 3);
    const int32_t* cells = table<int32_t>(hp, "ore_cells", &n);
    if (!st || !ci || !cf || !thr || !cells || n > 640 || !in_range(cells, n, 0, t.H * t.W) ||
        !in_range(st, 5, 1, t.nstates))
      return fail(MP_ERR_PACK, "mp_create: coop_mining tables missing");
    c.ore_cells = e->dev<int32_t>(cells); c.n_ore = (int)n;
    c.reward = e->dev<double>(cf);
    for (int k = 0; k < 3; ++k) c.thr[k] = thr[k];
    c.s_wait = st[0]; c.s_raw[0] = st[1]; c.s_raw[1] = st[2]; c.s_partial[0] = st[3]; c.s_partial[1] = st[4];
    c.cooldown = ci[0]; c.hit = ci[3]; c.ee_min_frames = ci[4]; c.ee_interval = ci[5];
    c.min_miners1 = ci[8]; c.window1 = ci[9];
    c.ore_layer = slayer[c.s_wait];
    // (type 0: extracted by the hit that mines it — one miner, no partial state of its own;
    // type 1's miners are a byte mask: the Lua's minNumMiners doubles as the type index)
    if (ci[6] != 1 || c.s_partial[0] != c.s_raw[0] || c.min_miners1 < 2 || c.min_miners1 > t.P_pack ||
        t.P_pack > 8 || c.window1 < 1 || c.window1 > 255 || c.cooldown < 1 || c.cooldown > 255 ||
        c.hit < 0 || c.hit >= e->nhits || c.ee_interval <= 0 || c.ore_layer < 0 ||
        c.ore_layer == t.avatar_layer || make_shape(ci[1], ci[2], &c.shape) > 16)
      return fail(MP_ERR_PACK, "mp_create: coop_mining constants out of engine range");
    for (int k = 1; k < 5; ++k)
      if (slayer[st[k]] != c.ore_layer)
        return fail(MP_ERR_PACK, "mp_create: coop_mining ore states on different layers");
    c.s_beam = hit_state[c.hit]; c.beam_layer = slayer[c.s_beam];
    if (c.beam_layer < 0 || c.beam_layer == c.ore_layer || c.beam_layer == t.avatar_layer)
      return fail(MP_ERR_PACK, "mp_create: coop_mining beam layer out of engine range");
    c.plane_m = t.L; c.plane_c = t.L + 1;
  }

  if (e->substrate == MPK_SUBSTRATE_COLLABORATIVE_COOKING) {
    CookTables& c = e->cc;
    const int32_t* st = table_n<int32_t>(hp, "cc_inv_states", 4);
    const int32_t* ci = table_n<int32_t>(hp, "cc_i32", 3);
    const double* cf = table_n<double>(hp, "cc_f64", 1);
    const int32_t* ps = table_n<int32_t>(hp, "cc_pot_states", 5);
    const int32_t* bs = table_n<int32_t>(hp, "cc_bar_states", 11);
    const int32_t* hits = table_n<int32_t>(hp, "cc_hits", (uint64_t)t.P_pack);
    const uint8_t* kind = table_n<uint8_t>(hp, "cc_state_kind", (uint64_t)t.nstates);
    uint64_t n_cont = 0, n_pot = 0, n_recv = 0, n_ci = 0, n_ri = 0, n_rf = 0;
    const int32_t* cont = table<int32_t>(hp, "cc_container_cells", &n_cont);
    const int32_t* cont_i = table<int32_t>(hp, "cc_container_i32", &n_ci);
    const int32_t* pots = table<int32_t>(hp, "cc_pot_cells", &n_pot);
    const int32_t* recv_i = table<int32_t>(hp, "cc_receiver_i32", &n_ri);
    const double* recv_f = table<double>(hp, "cc_receiver_f64", &n_rf);
    table<int32_t>(hp, "cc_receiver_cells", &n_recv);
    if (!st || !ci || !cf || !ps || !bs || !hits || !kind || (n_cont && (!cont || !cont_i)) ||
        (n_pot && !pots) || n_cont > 128 || n_pot > 64 || n_ci != 2 * n_cont || n_ri != 2 * n_recv ||
        n_rf != n_recv || (n_recv && (!recv_i || !recv_f)) ||
        !in_range(cont, n_cont, 0, t.H * t.W) || !in_range(pots, n_pot, 0, t.H * t.W) ||
        !in_range(ps, 5, 1, t.nstates) || !in_range(bs, 11, 1, t.nstates) || !in_range(st, 4, 1, t.nstates) ||
        !in_range(hits, (uint64_t)t.P_pack, 0, e->nhits))
      return fail(MP_ERR_PACK, "mp_create: collaborative_cooking tables missing");
    c.state_kind = e->dev<uint8_t>(kind);
    c.n_cont = (int)n_cont; c.n_pot = (int)n_pot;
    c.cont_cells = n_cont ? e->dev<int32_t>(cont) : nullptr;
    c.cont_i32 = n_cont ? e->dev<int32_t>(cont_i) : nullptr;
    c.pot_cells = n_pot ? e->dev<int32_t>(pots) : nullptr;
    for (int k = 0; k < 5; ++k) c.s_pot[k] = ps[k];
    c.s_bar0 = bs[0];
    c.s_plain0 = st[1]; c.s_off0 = st[2]; c.s_dir0 = st[3];
    c.overlay_layer = slayer[c.s_plain0];
    c.plane_t = t.L;
    c.cooldown = ci[0]; c.cooking_time = ci[1]; c.bar_interval = ci[2];
    c.pot_reward = cf[0];
    c.recv_item = n_recv ? recv_i[0] : -1; c.recv_global = n_recv ? recv_i[1] : 0;
    c.recv_reward = n_recv ? recv_f[0] : 0.0;
    c.s_beam0 = hit_state[hits[0]]; c.beam_layer0 = slayer[c.s_beam0];
    bool ok = c.s_plain0 + 4 <= t.nstates && c.s_off0 + 4 <= t.nstates && c.s_dir0 + 12 <= t.nstates &&
              c.overlay_layer >= 0 && c.overlay_layer != t.avatar_layer && c.cooldown <= 255 &&
              c.cooking_time >= 1 && c.cooking_time <= 30 && c.bar_interval >= 1;
    for (int k = 0; ok && k < 11; ++k) ok = bs[k] == bs[0] + k && slayer[bs[k]] == c.overlay_layer;
    for (int k = 0; ok && k < 4; ++k)
      ok = slayer[c.s_plain0 + k] == c.overlay_layer && slayer[c.s_off0 + k] == c.overlay_layer;
    for (int k = 0; ok && k < 12; ++k) ok = slayer[c.s_dir0 + k] == c.overlay_layer;
    for (int k = 0; ok && k < 5; ++k) ok = slayer[ps[k]] == t.avatar_layer;
    for (int p = 0; ok && p < t.P_pack; ++p)
      ok = hit_state[hits[p]] == c.s_beam0 + p && slayer[c.s_beam0 + p] == c.beam_layer0 + p &&
           c.beam_layer0 + p < t.L && c.beam_layer0 + p != c.overlay_layer && c.beam_layer0 + p != t.avatar_layer;
    for (uint64_t i = 0; ok && i < n_cont; ++i) ok = cont_i[2 * i] >= 0 && cont_i[2 * i] < 4;
    for (uint64_t i = 1; ok && i < n_recv; ++i)
      ok = recv_i[2 * i] == recv_i[0] && recv_i[2 * i + 1] == recv_i[1] && recv_f[i] == recv_f[0];
    if (!ok) return fail(MP_ERR_PACK, "mp_create: collaborative_cooking constants out of engine range");
  }

  if (e->substrate == MPK_SUBSTRATE_GIFT_REFINEMENTS) {
    GiftTables& c = e->gr;
    const int32_t* st = table_n<int32_t>(hp, "gr_states", 2);
    const int32_t* ci = table_n<int32_t>(hp, "gr_i32", 10);
    const double* cf = table_n<double>(hp, "gr_f64", 2 * (uint64_t)t.P_pack + 3);
    const uint64_t* thr = table_n<uint64_t>(hp, "gr_thr", 2);
    const int32_t* cells = table<int32_t>(hp, "token_cells", &n);
    if (!st || !ci || !cf || !thr || !cells || n > 640 || !in_range(cells, n, 0, t.H * t.W) ||
        !in_range(st, 2, 1, t.nstates))
      return fail(MP_ERR_PACK, "mp_create: gift_refinements tables missing");
    c.token_cells = e->dev<int32_t>(cells); c.n_token = (int)n;
    c.reward = e->dev<double>(cf);
    c.pick_reward = cf[2 * t.P_pack];
    c.thr[0] = thr[0]; c.thr[1] = thr[1];
    c.s_wait = st[0]; c.s_live = st[1];
    c.cooldown = ci[0]; c.hit = ci[3]; c.ee_min_frames = ci[4]; c.ee_interval = ci[5];
    c.capacity = ci[6]; c.ntypes = ci[7]; c.multiplier = ci[8]; c.consume_cooldown = ci[9];
    c.token_layer = slayer[c.s_live];
    // (an event row carries player | type << 4 and player | count << 4 in a byte each)
    if (c.capacity < 1 || c.capacity > 15 || c.ntypes < 1 || c.ntypes > 3 || c.multiplier < 1 ||
        c.multiplier > 255 || c.consume_cooldown < 0 || c.consume_cooldown > 255 || t.P_pack > 15 ||
        c.cooldown < 1 || c.cooldown > 255 || c.hit < 0 || c.hit >= e->nhits || c.ee_interval <= 0 ||
        c.token_layer < 0 || c.token_layer == t.avatar_layer || slayer[c.s_wait] != c.token_layer ||
        make_shape(ci[1], ci[2], &c.shape) > 16)
      return fail(MP_ERR_PACK, "mp_create: gift_refinements constants out of engine range");
    c.s_beam = hit_state[c.hit]; c.beam_layer = slayer[c.s_beam];
    if (c.beam_layer < 0 || c.beam_layer == c.token_layer || c.beam_layer == t.avatar_layer)
      return fail(MP_ERR_PACK, "mp_create: gift_refinements beam layer out of engine range");
  }

  if (e->substrate == MPK_SUBSTRATE_COINS) {
    CoinsTables& c = e->co;
    const int32_t* st = table_n<int32_t>(hp, "co_states", 3);
    const int32_t* ci = table_n<int32_t>(hp, "co_i32", 4);
    const double* cf = table_n<double>(hp, "co_f64", 8);
    const uint64_t* thr = table_n<uint64_t>(hp, "co_thr", 2);
    const int32_t* cells = table<int32_t>(hp, "coin_cells", &n);
    if (!st || !ci || !cf || !thr || !cells || n > 512 || t.P != 2 || t.P_pack != 2 ||
        !in_range(st, 3, 1, t.nstates) || !in_range(cells, n, 0, t.H * t.W))
      return fail(MP_ERR_PACK, "mp_create: coins tables missing or out of engine range");
    c.coin_cells = e->dev<int32_t>(cells); c.n_coin = (int)n;
    c.s_coin[0] = st[0]; c.s_coin[1] = st[1]; c.s_wait = st[2];
    c.coin_layer = slayer[st[0]]; c.wait_layer = slayer[st[2]];
    if (slayer[st[1]] != c.coin_layer || c.coin_layer < 0 || c.wait_layer < 0 ||
        c.coin_layer == t.avatar_layer)
      return fail(MP_ERR_PACK, "mp_create: coins layers out of engine range");
    for (int p = 0; p < t.P; ++p) {
      c.player_type[p] = ci[p];
      for (int k = 0; k < 4; ++k) c.rew[p][k] = cf[4 * p + k];
    }
    c.ee_min_frames = ci[t.P]; c.ee_interval = ci[t.P + 1];
    c.thr_regrow = thr[0]; c.thr_ee = thr[1];
    if (c.ee_interval <= 0) return fail(MP_ERR_PACK, "mp_create: coins constants out of range");
    const int32_t* cc = table_n<int32_t>(hp, "co_colour_coin", 5);
    const int32_t* ca = table_n<int32_t>(hp, "co_colour_alive", 10);
    c.has_colours = cc && ca;
    if (c.has_colours) {
      c.colour_coin = c.colour_alive[0] = c.colour_alive[1] = 0;
      for (int k = 0; k < 5; ++k) {
        c.colour_coin |= (uint64_t)(uint8_t)cc[k] << (8 * k);
        c.colour_alive[0] |= (uint64_t)(uint8_t)ca[k] << (8 * k);
        c.colour_alive[1] |= (uint64_t)(uint8_t)ca[5 + k] << (8 * k);
        if (cc[k] < 1 || cc[k] >= t.nstates || slayer[cc[k]] != c.coin_layer || ca[k] < 1 ||
            ca[k] >= t.nstates || ca[5 + k] < 1 || ca[5 + k] >= t.nstates ||
            slayer[ca[k]] != t.avatar_layer || slayer[ca[5 + k]] != t.avatar_layer)
          return fail(MP_ERR_PACK, "mp_create: coins colour tables out of range");
      }
    }
  }

  if (e->substrate == MPK_SUBSTRATE_CLEAN_UP) {
    CleanUpTables& c = e->cu;
    c.zap = zap;
    const int32_t* st = table_n<int32_t>(hp, "cu_states", 8);
    const int32_t* ci = table_n<int32_t>(hp, "cu_i32", 7);
    const double* cf = table_n<double>(hp, "cu_f64", 6);
    const uint64_t* misc = table_n<uint64_t>(hp, "thr_misc", 2);
    uint64_t na = 0, nd2 = 0, nw = 0;
    const int32_t* acells = table<int32_t>(hp, "apple_cells", &na);
    const int32_t* dcells = table<int32_t>(hp, "dirt_cells", &nd2);
    const int32_t* wcells = table<int32_t>(hp, "water_cells", &nw);
    const uint64_t* athr = table<uint64_t>(hp, "apple_thr", &n);
    c.clean_hit = find_name(hp, "hit_names", "cleanHit");
    if (!st || !ci || !cf || !misc || !acells || !dcells || !wcells || !athr ||
        n != nd2 + 1 || nd2 > 256 || na > 256 || nw > 256 || e->nhits != 2 || c.clean_hit < 0 ||
        !in_range(st, 8, 1, t.nstates) || !in_range(acells, na, 0, t.H * t.W) ||
        !in_range(dcells, nd2, 0, t.H * t.W) || !in_range(wcells, nw, 0, t.H * t.W))
      return fail(MP_ERR_PACK, "mp_create: clean_up tables missing or inconsistent");
    c.apple_cells = e->dev<int32_t>(acells); c.n_apple = (int)na;
    c.dirt_cells = e->dev<int32_t>(dcells); c.n_dirt = (int)nd2;
    c.water_cells = e->dev<int32_t>(wcells); c.n_water = (int)nw;
    c.apple_thr = e->dev<uint64_t>(athr);
    c.thr_dirt_spawn = misc[0]; c.thr_episode_end = misc[1];
    c.s_apple = st[0]; c.s_apple_wait = st[1]; c.s_dirt = st[2]; c.s_dirt_wait = st[3];
    c.s_water_packed = 0;
    for (int i = 0; i < 4; ++i) {
      c.s_water[i] = st[4 + i];
      c.s_water_packed |= (uint32_t)(st[4 + i] & 255) << (8 * i);
    }
    c.apple_layer = slayer[c.s_apple]; c.dirt_layer = slayer[c.s_dirt];
    c.dirt_wait_layer = slayer[c.s_dirt_wait]; c.water_layer = slayer[c.s_water[0]];
    c.s_clean_hit = hit_state[c.clean_hit];
    c.clean_layer = slayer[c.s_clean_hit];
    c.clean_cooldown = ci[0]; c.clean_length = ci[1]; c.clean_radius = ci[2];
    c.dirt_delay = ci[3]; c.ee_min_frames = ci[4]; c.ee_interval = ci[5];
    c.anim_frames = ci[6];
    c.eat_reward = cf[5];
    if (c.clean_cooldown > 255 || slayer[c.s_apple_wait] >= 0 ||
        c.apple_layer < 0 || c.dirt_layer < 0 || c.dirt_wait_layer < 0 || c.water_layer < 0 ||
        make_shape(c.clean_length, c.clean_radius, &c.clean_shape) > 16 ||
        !only_beams_on(c.clean_layer, c.s_clean_hit) || c.ee_interval <= 0 || c.anim_frames <= 0)
      return fail(MP_ERR_PACK, "mp_create: clean_up constants out of engine range");
    const uint8_t* ig = table<uint8_t>(hp, "init_grid");
    int nd = 0;
    for (int i = 0; i < t.H * t.W; ++i)
      nd += ig[c.dirt_layer * t.H * t.W + i] == c.s_dirt;
    c.n_dirt_init = nd;
  } else if (e->substrate == MPK_SUBSTRATE_COMMONS_HARVEST) {
    CommonsTables& c = e->ch;
    c.zap = zap;
    const int32_t* ci = table_n<int32_t>(hp, "ch_i32", 4);
    const int32_t* st = table_n<int32_t>(hp, "ch_states", ci && ci[0] > 0 && ci[0] <= 32 ? 4 + ci[0] : 4);
    const double* cf = table_n<double>(hp, "ch_f64", 1);
    const int32_t* cells = table<int32_t>(hp, "apple_cells", &n);
    if (!st || !ci || !cf || !cells || n > 256 || !in_range(cells, n, 0, t.H * t.W) ||
        !in_range(st, 4, 1, t.nstates))
      return fail(MP_ERR_PACK, "mp_create: commons_harvest tables missing");
    c.apple_cells = e->dev<int32_t>(cells); c.n_apple = (int)n;
    c.nk = ci[0]; c.ee_min_frames = ci[1]; c.ee_interval = ci[2];
    if (c.nk > 32 || c.nk < 1 || ci[3] != 1 || c.ee_interval <= 0)
      return fail(MP_ERR_PACK, "mp_create: commons_harvest constants out of engine range");
    c.s_apple = st[0]; c.s_wait = st[1]; c.s_grass = st[2]; c.s_dess = st[3];
    if (!in_range(st + 4, c.nk, 1, t.nstates))
      return fail(MP_ERR_PACK, "mp_create: appleWait_k states out of range");
    for (int k = 0; k < c.nk; ++k) c.s_wait_k[k] = st[4 + k];
    c.live_layer = slayer[c.s_apple]; c.wait_layer = slayer[c.s_wait];
    c.grass_layer = slayer[c.s_grass];
    c.eat_reward = cf[0];
    const int32_t* disc = table<int32_t>(hp, "disc_offsets", &n);
    c.disc = e->dev<int32_t>(disc); c.ndisc = (int)(n / 2);
    const uint64_t* thr = table<uint64_t>(hp, "ch_thr", &n);
    if (!disc || !thr || (int)n != c.nk + 1 || c.ndisc + 1 > c.nk || c.ndisc > 64 ||
        !in_range(disc, (uint64_t)c.ndisc * 2, -8, 9) ||
        c.live_layer < 0 || c.wait_layer < 0 || slayer[c.s_dess] != c.grass_layer)
      return fail(MP_ERR_PACK, "mp_create: commons_harvest tables inconsistent");
    for (int k = 0; k < c.nk; ++k)
      if (slayer[c.s_wait_k[k]] != c.wait_layer)
        return fail(MP_ERR_PACK, "mp_create: appleWait_k states on different layers");
    c.thr = e->dev<uint64_t>(thr);
  } else if (e->substrate == MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS) {
    MushroomTables& c = e->em;
    c.zap = zap;
    const int32_t* st = table_n<int32_t>(hp, "em_states", 8);
    const int32_t* ci = table_n<int32_t>(hp, "em_i32", 30);
    const double* cf = table_n<double>(hp, "em_f64", 8);
    const uint64_t* thr = table_n<uint64_t>(hp, "em_thr", 21);
    const int32_t* cells = table<int32_t>(hp, "mushroom_cells", &n);
    if (!st || !ci || !cf || !thr || !cells || n < 1 || n > 256 ||   // 4 per lane, step_mushroom.h
        !in_range(st, 8, 1, t.nstates) || !in_range(cells, n, 0, t.H * t.W))
      return fail(MP_ERR_PACK, "mp_create: externality_mushrooms tables missing");
    c.site_cells = e->dev<int32_t>(cells); c.n_site = (int)n;
    c.i32 = e->dev<int32_t>(ci); c.thr = e->dev<uint64_t>(thr);
    c.s_type0 = st[0]; c.live_layer = slayer[st[0]];
    c.s_mark[0] = st[5]; c.s_mark[1] = st[6]; c.mark_layer = slayer[st[5]];
    c.plane_age = t.L;
    c.min_potential = ci[0]; c.recovery_time = ci[2];
    c.ee_min_frames = ci[4]; c.ee_interval = ci[5]; c.n_live_init = ci[7];
    bool ok = st[1] == st[0] + 1 && st[2] == st[0] + 2 && st[3] == st[0] + 3 &&
              slayer[st[4]] < 0 && slayer[st[7]] < 0 && slayer[st[6]] == c.mark_layer &&
              c.live_layer >= 0 && c.mark_layer >= 0 && c.live_layer != t.avatar_layer &&
              c.mark_layer != t.avatar_layer && c.live_layer != c.mark_layer &&
              ci[1] == 1 && ci[3] == 2 && ci[6] == zap.hit && c.ee_interval > 0 &&
              c.recovery_time >= 1 && c.recovery_time <= 255 && c.n_live_init >= 0 &&
              c.n_live_init <= (int)n && !zap.remove_hit && zap.penalty == 0.0 && zap.reward == 0.0 &&
              zap.respawn_frames >= 1 && t.n_optional == 0 && t.P >= 2;
    // the mushrooms' plane and the markings' hold nothing else, every mushroom site starts
    // on the map as the object table says
    for (int s = 1; s < t.nstates && ok; ++s) {
      if (slayer[s] == c.live_layer && (s < st[0] || s > st[3])) ok = false;
      if (slayer[s] == c.mark_layer && s != st[5] && s != st[6]) ok = false;
    }
    c.perish_packed = 0;
    for (int k = 0; k < 4 && ok; ++k) {
      const int delay = ci[16 + k];   // (the age plane saturates at 255)
      ok = ci[8 + k] >= 0 && ci[8 + k] <= 4 && ci[12 + k] >= 0 && ci[12 + k] <= 255 &&
           delay >= 1 && (delay <= 254 || delay >= (1 << 30)) && ci[20 + k] >= -1 && ci[20 + k] < 4;
      c.perish_packed |= (uint32_t)(delay <= 254 ? delay : 255) << (8 * k);
    }
    if (!ok) return fail(MP_ERR_PACK, "mp_create: externality_mushrooms constants out of engine range");
    for (int l = 0; l < 2; ++l) {
      c.lv_increment[l] = ci[24 + 3 * l]; c.lv_freeze[l] = ci[25 + 3 * l];
      c.lv_remove[l] = ci[26 + 3 * l];
      c.lv_source[l] = cf[4 + 2 * l]; c.lv_target[l] = cf[5 + 2 * l];
      if (c.lv_freeze[l] < 0 || c.lv_freeze[l] > 255 || c.lv_increment[l] < -1 || c.lv_increment[l] > 1)
        return fail(MP_ERR_PACK, "mp_create: externality_mushrooms sanction levels out of engine range");
    }
    // _rewardEveryone (components.lua:65-105) with this engine's player count
    c.pays = 0;
    for (int k = 0; k < 4; ++k) { c.rew_self[k] = 0.0; c.rew_other[k] = 0.0; }
    c.rew_self[0] = cf[0]; c.pays |= 1u;
    c.rew_self[1] = c.rew_other[1] = cf[1] / (double)t.P; c.pays |= (1u << 1) | (1u << 5);
    c.rew_other[2] = cf[2] / (double)(t.P - 1); c.pays |= 1u << 6;
    c.rew_self[3] = c.rew_other[3] = cf[3] / (double)t.P; c.pays |= (1u << 3) | (1u << 7);
    c.thr_ee = thr[20];
  } else if (e->substrate == MPK_SUBSTRATE_TERRITORY) {
    TerritoryTables& c = e->tr;
    c.zap = zap;
    const int32_t* st = table_n<int32_t>(hp, "tr_states", 10 + 2 * (uint64_t)t.P_pack);
    const int32_t* ci = table_n<int32_t>(hp, "tr_i32", 16);
    const double* cf = table_n<double>(hp, "tr_f64", 8);
    const uint64_t* thr = table_n<uint64_t>(hp, "tr_thr", 3);
    const int32_t* hits = table_n<int32_t>(hp, "tr_hits", 1 + 2 * (uint64_t)t.P_pack);
    const int32_t* hsd = table<int32_t>(hp, "hit_state_dir");
    const int32_t* cells = table<int32_t>(hp, "resource_cells", &n);
    if (!st || !ci || !cf || !thr || !hits || !cells || n > 256 ||   // 4 per lane, step_territory.h
        !in_range(st, 10 + 2 * (uint64_t)t.P_pack, 1, t.nstates) ||
        !in_range(hits, 1 + 2 * (uint64_t)t.P_pack, 0, e->nhits) ||
        !in_range(cells, n, 0, t.H * t.W))
      return fail(MP_ERR_PACK, "mp_create: territory tables missing");
    c.res_cells = e->dev<int32_t>(cells); c.n_res = (int)n;
    c.map_cells = t.H * t.W;
    const int P = t.P_pack;   // table strides; absent players' states are never on the grid
    c.s_res_unclaimed = st[0]; c.s_dmg_inactive = st[5]; c.s_dmg_damaged = st[6];
    c.s_mark[0] = st[7]; c.s_mark[1] = st[8];
    for (int p = 0; p < P; ++p) { c.s_claimed[p] = st[10 + p]; c.s_dry[p] = st[10 + P + p]; }
    c.res_layer = slayer[st[0]]; c.tex_layer = slayer[st[2]];
    c.ind_layer = slayer[c.s_dry[0]]; c.dmg_layer = slayer[st[5]]; c.mark_layer = slayer[st[7]];
    c.plane_a = t.L; c.plane_b = t.L + 1; c.plane_c = t.L + 2;
    c.initial_health = ci[0]; c.reward_delay = ci[1]; c.repair_delay = ci[2];
    c.claim_length = ci[3]; c.claim_wait = ci[5]; c.recovery_time = ci[6];
    c.ee_min_frames = ci[8]; c.ee_interval = ci[9];
    if (ci[7] != 2 || ci[4] != 0 || c.initial_health > 3 || c.claim_length < 1 ||
        c.claim_length * t.P > 64 || c.ee_interval <= 0 || zap.remove_hit || c.res_layer != t.avatar_layer ||
        slayer[st[1]] >= 0 || slayer[st[3]] >= 0 || slayer[st[4]] >= 0 || slayer[st[9]] >= 0)
      return fail(MP_ERR_PACK, "mp_create: territory constants out of engine range");
    for (int l = 0; l < 2; ++l) {
      c.lv_increment[l] = ci[10 + 3 * l]; c.lv_freeze[l] = ci[11 + 3 * l];
      c.lv_remove[l] = ci[12 + 3 * l];
      c.lv_source[l] = cf[4 + 2 * l]; c.lv_target[l] = cf[5 + 2 * l];
      if (c.lv_freeze[l] > 255) return fail(MP_ERR_PACK, "mp_create: freeze too long");
    }
    c.reward = cf[0];
    c.thr_reward = thr[0]; c.thr_repair = thr[1]; c.thr_ee = thr[2];
    c.hit_zap = hits[0];
    for (int p = 0; p < P; ++p) {
      c.hit_brush[p] = hits[1 + p]; c.hit_claim[p] = hits[1 + P + p];
      for (int d = 0; d < 4; ++d) c.s_brush[p][d] = hsd[c.hit_brush[p] * 4 + d];
      c.s_claim_hit[p] = hit_state[c.hit_claim[p]];
    }
    c.brush_layer = slayer[c.s_brush[0][0]]; c.claim_layer = slayer[c.s_claim_hit[0]];
    for (int s = 1; s < t.nstates; ++s) {  // hit layers hold nothing but beam sprites
      bool is_hit = false;
      for (int h = 0; h < e->nhits * 4; ++h) is_hit = is_hit || hsd[h] == s;
      if (!is_hit && (slayer[s] == c.brush_layer || slayer[s] == c.claim_layer))
        return fail(MP_ERR_PACK, "mp_create: a piece state lives on a territory hit layer");
    }
  }

  const size_t state_bytes = (size_t)e->N * t.world_stride;
  DEV_ALLOC(e->d_state, state_bytes);
  {
    std::vector<uint8_t> init(state_bytes, 0);
    for (int w = 0; w < e->N; ++w) {
      WorldTail* tail = reinterpret_cast<WorldTail*>(
          init.data() + (size_t)w * t.world_stride + t.grid_pad);
      const uint64_t gw = cfg->world_offset + (uint64_t)w;
      tail->seed = (cfg->base_seed || cfg->literal_base_seed) ? cfg->base_seed + gw
                                                              : 0x9E3779B97F4A7C15ull * (gw + 1);
    }
    HIP_TRY(hipMemcpy(e->d_state, init.data(), state_bytes, hipMemcpyHostToDevice));
  }
  {
    const size_t NP = (size_t)e->N * t.P, N = (size_t)e->N;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_reward = take(NP * 8), o_ready = take(NP * 8), o_aux = take(NP * 8),
                 o_disc = take(N * 8), o_coll = take(N * 8), o_type = take(N * 4),
                 o_pos = take(NP * 8), o_ori = take(NP * 4),
                 o_ev = take(N * MP_EVENT_ROWS * 16);
    const bool matrix = e->substrate == MPK_SUBSTRATE_THE_MATRIX;
    const size_t o_inv = take(NP * e->inventory_types() * 8),
                 o_int = take(matrix ? NP * 2 * e->mx.R * 8 : 0),
                 o_irw = take(matrix ? NP * 2 * 8 : 0);
    DEV_ALLOC(e->d_scalars, off);
    e->scalars_bytes = off;
    HIP_TRY(hipMemset(e->d_scalars, 0, off));
    e->own.reward = (double*)(e->d_scalars + o_reward);
    e->own.ready = (double*)(e->d_scalars + o_ready);
    e->own.aux0 = (double*)(e->d_scalars + o_aux);
    e->own.discount = (double*)(e->d_scalars + o_disc);
    e->own.collective = (double*)(e->d_scalars + o_coll);
    e->own.step_type = (int32_t*)(e->d_scalars + o_type);
    e->own.position = (int32_t*)(e->d_scalars + o_pos);
    e->own.orientation = (int32_t*)(e->d_scalars + o_ori);
    e->own.events = (int32_t*)(e->d_scalars + o_ev);
    if (e->inventory_types() > 0) e->own.inventory = (double*)(e->d_scalars + o_inv);
    if (matrix) {
      e->own.interaction = (double*)(e->d_scalars + o_int);
      e->own.interaction_rewards = (double*)(e->d_scalars + o_irw);
    }
    if (cfg->debug_observations) {
      size_t doff = 0;
      auto dtake = [&](size_t bytes) { size_t o = doff; doff += (bytes + 255) & ~(size_t)255; return o; };
      size_t o_dbg[4];
      for (int k = 0; k < 4; ++k) o_dbg[k] = dtake(NP * 8);
      const size_t o_zm = dtake(NP * t.P * 8);
      const size_t o_cum = dtake(matrix ? NP * (1 + 3 * e->mx.R) * 8 : 0);
      DEV_ALLOC(e->d_debug, doff);
      e->debug_bytes = doff;
      HIP_TRY(hipMemset(e->d_debug, 0, doff));
      if (e->substrate == MPK_SUBSTRATE_CLEAN_UP)
        for (int k = 0; k < 4; ++k) e->own.dbg[k] = (double*)(e->d_debug + o_dbg[k]);
      if (e->substrate == MPK_SUBSTRATE_CLEAN_UP || e->substrate == MPK_SUBSTRATE_COMMONS_HARVEST)
        e->own.zap_matrix = (double*)(e->d_debug + o_zm);
      if (matrix) e->own.cumulants = (double*)(e->d_debug + o_cum);
    }
    DEV_ALLOC(e->d_actions, NP * 4);
    DEV_ALLOC(e->d_mask, N);
    DEV_ALLOC(e->d_seeds, N * 8);
    DEV_ALLOC(e->d_ctr, MP_CTR_COUNT * 8);
  }
#undef DEV_ALLOC
  // renderer: de-duplicated sprite atlas (noRotate sprites and solid colours
  // have four identical facings), opaque sprites stored with alpha cleared
  {
    const uint8_t* rgba = table<uint8_t>(hp, "sprite_rgba");
    const int32_t* flags = table<int32_t>(hp, "sprite_flags");
    const int nimg = t.nsprites * 4;
    std::vector<uint8_t> images(256, 0);  // image 0: unused padding
    std::vector<uint16_t> slots((size_t)nimg, 0);
    int count = 1;
    for (int i = 0; i < nimg; ++i) {
      uint8_t img[256];
      memcpy(img, rgba + (size_t)i * 256, 256);
      if (flags[i >> 2] & MPK_SPRITE_OPAQUE) {
        // opaque images are only ever copied: store them pre-packed, 8 rows of
        // 24 B RGB followed by 8 B of padding
        uint8_t packed[256] = {0};
        for (int py = 0; py < 8; ++py)
          for (int px = 0; px < 8; ++px)
            for (int ch = 0; ch < 3; ++ch)
              packed[py * 32 + px * 3 + ch] = img[(py * 8 + px) * 4 + ch];
        memcpy(img, packed, 256);
      }
      int found = -1;
      for (int k = 1; k < count && found < 0; ++k)
        if (memcmp(images.data() + (size_t)k * 256, img, 256) == 0) found = k;
      if (found < 0) {
        found = count++;
        images.insert(images.end(), img, img + 256);
      }
      slots[(size_t)i] = (uint16_t)found;
    }
    // composite cache (render.hip phase 1): pre-blend the (opaque base, overlay)
    // stacks that the map's static pieces can form — dirt on water, shadows on
    // sand, claimed-resource paint on its texture ... — so such cells become plain
    // copies.  A piece's possible looks are all sprite-bearing states of its
    // prefab ("prefab.state" names); avatars, their markings and beams move, so
    // they are never part of a cached stack.
    t.scratch_cells = (dev && dev->scratch_cells > 0) ? dev->scratch_cells : 8;
    std::vector<uint32_t> pair_table(kPairSlots, 0xffffffffu);
    int pair_probe = 0, n_composites = 0, used_slots = 0;
    if (!(dev && dev->no_composite_cache)) {
      uint64_t names_len = 0;
      const char* names = table<char>(hp, "state_names", &names_len);
      const int32_t* objs = table<int32_t>(hp, "objects");
      const int32_t* st_layer = table<int32_t>(hp, "state_layer");
      const int32_t* st_sprite = table<int32_t>(hp, "state_sprite");
      const int32_t* st_orient = table<int32_t>(hp, "state_orient");
      const int nobj = hdr[MPK_HDR_NOBJ];
      std::vector<std::string> prefab((size_t)t.nstates);
      {
        uint64_t off = 0;
        for (int s = 0; s < t.nstates && off < names_len; ++s) {
          const std::string nm(names + off);
          off += nm.size() + 1;
          prefab[(size_t)s] = nm.substr(0, nm.find('.'));
        }
      }
      struct Look { int layer, sprite, orient; };
      std::vector<std::vector<Look>> cell_looks((size_t)t.H * t.W);
      for (int i = 0; i < nobj; ++i) {
        const int32_t* ob = objs + 4 * i;
        if (ob[0] == MPK_KIND_SCENE || ob[0] == MPK_KIND_AVATAR || ob[0] == MPK_KIND_MARKING) continue;
        for (int s = 1; s < t.nstates; ++s)
          if (prefab[(size_t)s] == prefab[(size_t)ob[3]] && st_sprite[s] >= 0 && st_layer[s] >= 0)
            cell_looks[(size_t)ob[2] * t.W + ob[1]].push_back({st_layer[s], st_sprite[s], st_orient[s]});
      }
      // stacks: an opaque look, then up to two non-opaque looks on higher layers
      struct Stack { int n; Look l[3]; long cells; };
      std::vector<Stack> stacks;
      auto same = [](const Look& a, const Look& b) {
        return a.sprite == b.sprite && a.orient == b.orient;
      };
      auto count_stack = [&](const Look* l, int n) {
        for (auto& sk : stacks) {
          bool eq = sk.n == n;
          for (int k = 0; eq && k < n; ++k) eq = same(sk.l[k], l[k]);
          if (eq) { sk.cells++; return; }
        }
        Stack sk; sk.n = n; sk.cells = 1;
        for (int k = 0; k < n; ++k) sk.l[k] = l[k];
        stacks.push_back(sk);
      };
      auto overlay = [&](const Look& l) {
        return !(flags[l.sprite] & (MPK_SPRITE_OPAQUE | MPK_SPRITE_EMPTY));
      };
      for (const auto& looks : cell_looks) {
        // (sprite -1: no opaque piece below — the renderer starts from image 0,
        // black; the *_in_the_matrix maps have no floor under their resources)
        std::vector<Look> bases;
        for (const Look& a : looks)
          if (flags[a.sprite] & MPK_SPRITE_OPAQUE) bases.push_back(a);
        if (bases.empty()) bases.push_back({-1, -1, 0});   // a cell no piece can cover
        for (const Look& a : bases) {
          for (const Look& b : looks) {
            if (b.layer <= a.layer || !overlay(b)) continue;
            const Look ab[3] = {a, b, b};
            count_stack(ab, 2);
            for (const Look& c : looks) {
              if (c.layer <= b.layer || !overlay(c)) continue;
              const Look abc[3] = {a, b, c};
              count_stack(abc, 3);
            }
          }
        }
      }
      // stacks the lowering knows to be the common ones (territory: texture + wet +
      // dry paint of the SAME player, 9 of 81 combinations): first in their class
      {
        uint64_t nh = 0;
        const int32_t* hints = table<int32_t>(hp, "composite_hints", &nh);
        for (uint64_t i = 0; hints && i + 2 < nh; i += 3) {
          Look l[3]; int n = 0; bool ok = true;
          for (int k = 0; k < 3; ++k) {
            const int st = hints[i + k];
            if (st == 0 && k > 0) break;
            if (st <= 0 || st >= t.nstates || st_sprite[st] < 0 || st_layer[st] < 0) { ok = false; break; }
            l[n++] = {st_layer[st], st_sprite[st], st_orient[st]};
          }
          if (!ok || n < 2 || !(flags[l[0].sprite] & MPK_SPRITE_OPAQUE)) continue;
          for (int m = 2; m <= n; ++m) {
            count_stack(l, m);
            for (auto& sk : stacks) {
              bool eq = sk.n == m;
              for (int k = 0; eq && k < m; ++k) eq = same(sk.l[k], l[k]);
              if (eq) sk.cells = 1L << 40;
            }
          }
        }
      }
      std::sort(stacks.begin(), stacks.end(), [](const Stack& x, const Stack& y) {
        return x.n != y.n ? x.n < y.n : x.cells > y.cells;   // all pairs before triples
      });
      auto add_image = [&](const uint8_t* img) {
        for (int k = 0; k < count; ++k)
          if (memcmp(images.data() + (size_t)k * 256, img, 256) == 0) return k;
        images.insert(images.end(), img, img + 256);
        return count++;
      };
      auto lookup = [&](uint32_t a, uint32_t b) -> int {
        for (uint32_t h = pair_hash(a, b), k = 0; k < (uint32_t)kPairSlots; ++k) {
          const uint32_t ent = pair_table[(h + k) & (kPairSlots - 1)];
          if (ent == 0xffffffffu) return -1;
          if ((ent >> 10) == ((a << 10) | b)) return (int)(ent & 1023u);
        }
        return -1;
      };
      auto insert = [&](uint32_t a, uint32_t b, uint32_t c) {
        for (uint32_t h = pair_hash(a, b), k = 0; k < (uint32_t)kPairSlots; ++k) {
          uint32_t& ent = pair_table[(h + k) & (kPairSlots - 1)];
          if (ent == 0xffffffffu) {
            ent = (a << 20) | (b << 10) | c;
            if ((int)k + 1 > pair_probe) pair_probe = (int)k + 1;
            return;
          }
        }
      };
      // one overlay image blended onto a packed opaque image, exactly as
      // render.hip does it (A7: (s*a + d*(255-a) + 127) / 255; binary sprites
      // replace where alpha > 0)
      auto blend = [&](int base_img, int ov_img, bool partial, uint8_t* out) {
        memcpy(out, images.data() + (size_t)base_img * 256, 256);
        const uint8_t* ov = images.data() + (size_t)ov_img * 256;
        for (int py = 0; py < 8; ++py)
          for (int px = 0; px < 8; ++px) {
            const uint8_t* s = ov + (py * 8 + px) * 4;
            uint8_t* d = out + py * 32 + px * 3;
            const unsigned a = s[3];
            for (int ch = 0; ch < 3; ++ch) {
              if (partial) d[ch] = (uint8_t)((s[ch] * a + d[ch] * (255u - a) + 127u) / 255u);
              else if (a) d[ch] = s[ch];
            }
          }
      };
      // budget: whatever LDS the renderer's preferred geometry leaves free (more
      // images must not cost worlds per workgroup: measured, tools/sweep_env.sh)
      t.n_images = count;
      int kMaxComposites = kPairSlots;
      for (int v = 0; v < 6; ++v) {
        const FramePlan p0 = plan_frame(t, e->sub, e->N, (v & 1) != 0, v >> 1, e->num_cus, dev);
        kMaxComposites = std::min(kMaxComposites, (160 * 1024 - frame_lds_bytes(t, p0)) / 272);
      }
      if (kMaxComposites < 0) kMaxComposites = 0;
      if (kMaxComposites > kPairSlots / 2) kMaxComposites = kPairSlots / 2;
      if (dev && dev->max_composites >= 0)
        kMaxComposites = std::min(kMaxComposites, (int)dev->max_composites);
      for (const Stack& sk : stacks) {
        for (int f = 0; f < 4; ++f) {
          int base = sk.l[0].sprite < 0 ? 0 : slots[(size_t)sk.l[0].sprite * 4 + ((f + sk.l[0].orient) & 3)];
          bool ok = true;
          for (int k = 1; k < sk.n && ok; ++k) {
            const int ov = slots[(size_t)sk.l[k].sprite * 4 + ((f + sk.l[k].orient) & 3)];
            int comp = lookup((uint32_t)base, (uint32_t)ov);
            if (comp < 0) {
              // (a triple extends a cached pair; it is skipped if its pair was)
              if (k < sk.n - 1 || n_composites >= kMaxComposites || used_slots >= kPairSlots / 2 ||
                  count >= 1023) { ok = false; break; }
              uint8_t img[256];
              blend(base, ov, (flags[sk.l[k].sprite] & MPK_SPRITE_PARTIAL) != 0, img);
              const int before = count;
              comp = add_image(img);
              n_composites += count - before;
              insert((uint32_t)base, (uint32_t)ov, (uint32_t)comp);
              ++used_slots;
            }
            base = comp;
          }
        }
      }
    }
    if (count > 1023) return fail(MP_ERR_PACK, "mp_create: %d distinct sprite images", count);
    t.n_images = count;
    t.pair_probe = pair_probe;
    const size_t img_bytes = (size_t)count * 256, slot_bytes = ((size_t)nimg * 2 + 15) & ~(size_t)15,
                 pair_bytes = (size_t)kPairSlots * 4, blob_bytes = (size_t)render_blob_bytes(t);
    // what every render workgroup stages besides its worlds, already in LDS layout
    std::vector<uint8_t> blob(blob_bytes);
    {
      const int32_t* alive = table<int32_t>(hp, "avatar_alive_state");
      std::vector<uint8_t> flags8((size_t)t.nsprites);
      for (int s = 0; s < t.nsprites; ++s)
        flags8[(size_t)s] = (uint8_t)(((flags[s] & MPK_SPRITE_OPAQUE) ? 1 : 0) |
                                      ((flags[s] & MPK_SPRITE_PARTIAL) ? 2 : 0) |
                                      ((flags[s] & MPK_SPRITE_EMPTY) ? 4 : 0));
      std::vector<int8_t> splayer(256, -1);
      for (int p = 0; p < t.P; ++p) splayer[(size_t)alive[p]] = (int8_t)p;
      {
        uint64_t nx = 0;
        const int32_t* xa = table<int32_t>(hp, "avatar_extra_alive", &nx);
        for (uint64_t i = 0; xa && i + 1 < nx; i += 2)
          if (xa[i + 1] < t.P) splayer[(size_t)xa[i]] = (int8_t)xa[i + 1];
      }
      // viewers 0 .. P-1, then the world view (row P_pack of the pack's table)
      const int32_t* vmap = table<int32_t>(hp, "view_sprite_map");
      std::vector<int32_t> vmap_p((size_t)(t.P + 1) * t.nsprites);
      for (int v = 0; v <= t.P; ++v)
        memcpy(vmap_p.data() + (size_t)v * t.nsprites,
               vmap + (size_t)(v < t.P ? v : t.P_pack) * t.nsprites, (size_t)t.nsprites * 4);
      build_render_blob(t, images.data(), slots.data(), pair_table.data(),
                        table<int32_t>(hp, "state_sprite"), splayer.data(),
                        vmap_p.data(), flags8.data(),
                        table<int32_t>(hp, "state_orient"), blob.data());
      t.vis_layers = render_visible_layers(t, blob.data(), table<int32_t>(hp, "state_layer"));
      // (the renderers carry a plane's byte offset in a record as 16 bits: FrameConsts::plane_off)
      if ((t.L - 1) * t.H * t.W >= 65536)
        return fail(MP_ERR_PACK, "mp_create: %d render planes of %d x %d cells", t.L, t.H, t.W);
    }
    HIP_TRY(hipMalloc((void**)&e->d_atlas, img_bytes + slot_bytes + pair_bytes + blob_bytes));
    HIP_TRY(hipMemcpy(e->d_atlas, images.data(), img_bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_atlas + img_bytes, slots.data(), (size_t)nimg * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_atlas + img_bytes + slot_bytes, pair_table.data(), pair_bytes,
                      hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_atlas + img_bytes + slot_bytes + pair_bytes, blob.data(), blob_bytes,
                      hipMemcpyHostToDevice));
    t.atlas_compact = e->d_atlas;
    t.img_slot = reinterpret_cast<const uint16_t*>(e->d_atlas + img_bytes);
    t.pair_table = reinterpret_cast<const uint32_t*>(e->d_atlas + img_bytes + slot_bytes);
    t.render_blob = e->d_atlas + img_bytes + slot_bytes + pair_bytes;

    if (dev && dev->verbose)
      fprintf(stderr, "mp_engine: composite cache: %d images, %d table entries, probe %d\n",
              n_composites, used_slots, pair_probe);
    for (int v = 0; v < 6; ++v) {
      FramePlan& pl = e->plan[v & 1][v >> 1];
      pl = plan_frame(t, e->sub, e->N, (v & 1) != 0, v >> 1, e->num_cus, dev);
      if (frame_lds_bytes(t, pl) > 160 * 1024)
        return fail(MP_ERR_PACK, "mp_create: renderer needs %d B of LDS", frame_lds_bytes(t, pl));
    }
    if (int rc = prepare_frame())
      return fail(MP_ERR_HIP, "mp_create: hipFuncSetAttribute(max dynamic LDS) failed: %d", rc);
    if (dev && dev->verbose)
      for (int v = 0; v < 6; ++v) {
        const FramePlan& pl = e->plan[v & 1][v >> 1];
        fprintf(stderr, "mp_engine: %d sprite images; frame plan %s, %s: %d buffers x %d worlds, %d of %d waves feed"
                " (%d draw the world view), %d groups own %d batches each + %d pooled, %d B LDS\n",
                count, (v & 1) ? "stepping + drawing" : "drawing",
                (v >> 1) == 0 ? "agents view" : (v >> 1) == 1 ? "world view" : "both views",
                pl.NB, pl.B, pl.feeders, pl.nwaves, pl.world_waves, pl.groups, pl.ks, pl.pool,
                frame_lds_bytes(t, pl));
      }
  }
  return MP_OK;
}

void mp_destroy(MpEngine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  if (e->h_fault) (void)hipHostFree(e->h_fault);
  void* bufs[] = {e->d_pack, e->d_extra, e->d_stepblob, e->d_debug, e->d_state, e->d_scalars,
                  e->d_actions, e->d_fields, e->d_mask, e->d_seeds, e->d_atlas, e->d_ctr, e->d_claim};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  for (int i = 0; i < MpEngine::kHostSlots; ++i)
    if (e->h_actions[i]) { (void)hipHostFree(e->h_actions[i]); (void)hipEventDestroy(e->h_copied[i]); }
  delete e;
}

int mp_info(const MpEngine* e, MpInfo* out) {
  if (!e || !out) return fail(MP_ERR_INVALID, "mp_info: NULL argument");
  memset(out, 0, sizeof *out);
  out->abi_version = MP_ABI_VERSION;
  out->substrate = e->substrate;
  out->num_worlds = e->N; out->num_players = e->t.P; out->num_actions = e->t.nact;
  out->map_h = e->t.H; out->map_w = e->t.W; out->num_layers = e->t.L;
  out->sprite_size = e->t.sprite_size;
  out->view_h = e->t.vf + e->t.vb + 1; out->view_w = e->t.vl + e->t.vr + 1;
  out->max_frames = e->t.max_frames;
  out->world_state_bytes = e->t.world_stride;
  // the launch form of a step with the views bound right now (the per-agent view
  // if none is)
  out->fused = e->fuse(!e->bound[MP_OBS_RGB] && e->bound[MP_OBS_WORLD_RGB]) ? 1 : 0;
  out->num_resources = e->inventory_types();
  out->num_action_fields = e->t.nfields;
  {
    const bool a = e->bound[MP_OBS_RGB] != nullptr, w = e->bound[MP_OBS_WORLD_RGB] != nullptr;
    const int views = a && w ? 2 : w ? 1 : 0;
    // (with a tuned ring: the plan of the slot the next submission writes)
    const FramePlan& p = e->ring_slots > 0 && e->ring_plan[views].size() == (size_t)e->ring_slots
                             ? e->ring_plan[views][(size_t)(e->ring_cursor % (uint64_t)e->ring_slots)]
                             : e->plan[1][views];
    out->plan_batch_worlds = p.B; out->plan_ring_batches = p.NB; out->plan_owned_batches = p.ks;
    out->plan_pooled_batches = p.pool; out->plan_groups = p.groups;
    out->plan_store_sc1 = p.store_sc1;
    out->plan_feeders = p.feeders; out->plan_waves = p.nwaves;
    out->plan_pace = p.pace;
    out->plan_team = p.team;
    out->plan_late_priority = p.late_prio;
  }
  out->visible_layers = (int32_t)(e->t.vis_layers & 0xffffu);
  out->ring_slots = e->ring_slots;
  out->ring_next = e->ring_slots > 0 ? (int32_t)(e->ring_cursor % (uint64_t)e->ring_slots) : 0;
  retired_va(&out->retired_va_bytes, &out->retired_va_limit);
  return MP_OK;
}

int mp_set_stream(MpEngine* e, void* stream) {
  if (!e) return fail(MP_ERR_INVALID, "mp_set_stream: NULL engine");
  if ((hipStream_t)stream == e->stream) return MP_OK;
  // work already enqueued on the old stream is ordered before what follows on
  // the new one
  HIP_TRY(hipSetDevice(e->device));
  hipEvent_t ev;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t rc = hipEventRecord(ev, e->stream);
  if (rc == hipSuccess) rc = hipStreamWaitEvent((hipStream_t)stream, ev, 0);
  (void)hipEventDestroy(ev);
  HIP_TRY(rc);
  e->stream = (hipStream_t)stream;
  return MP_OK;
}

int mp_bind_output(MpEngine* e, MpObsKind kind, void* device_ptr) {
  if (!e || kind < 0 || kind >= MP_OBS_KINDS)
    return fail(MP_ERR_INVALID, "mp_bind_output: bad argument");
  if (device_ptr && mp_obs_bytes(e, kind) == 0)
    return fail(MP_ERR_UNSUPPORTED, "mp_bind_output: this substrate has no observation %d", (int)kind);
  if (device_ptr)
    if (int rc = check_device_pointer(e, device_ptr, "mp_bind_output")) return rc;
  e->bound[kind] = device_ptr;
  drop_ring_kind(e, kind);
  return MP_OK;
}

int mp_bind_output_ring(MpEngine* e, MpObsKind kind, void* base, uint64_t slot_stride_bytes,
                        int32_t slots) {
  if (!e || kind < 0 || kind >= MP_OBS_KINDS)
    return fail(MP_ERR_INVALID, "mp_bind_output_ring: bad argument");
  if (!base) return mp_bind_output(e, kind, nullptr);
  const uint64_t bytes = mp_obs_bytes(e, kind);
  if (bytes == 0)
    return fail(MP_ERR_UNSUPPORTED, "mp_bind_output_ring: this substrate has no observation %d", (int)kind);
  if (kind == MP_OBS_LAYER)
    return fail(MP_ERR_UNSUPPORTED, "mp_bind_output_ring: N.LAYER is not offered as a ring");
  if (slots < 1 || slots > (1 << 20))
    return fail(MP_ERR_INVALID, "mp_bind_output_ring: %d slots", (int)slots);
  if (slot_stride_bytes < bytes || (slot_stride_bytes & 255) != 0)
    return fail(MP_ERR_INVALID, "mp_bind_output_ring: a slot stride of %llu bytes for an observation of %llu "
                "(must hold it and be a multiple of 256)", (unsigned long long)slot_stride_bytes,
                (unsigned long long)bytes);
  if (int rc = check_device_pointer(e, base, "mp_bind_output_ring")) return rc;
  if (int rc = check_device_pointer(e, (const char*)base + (uint64_t)(slots - 1) * slot_stride_bytes + bytes - 1,
                                    "mp_bind_output_ring (last byte of the last slot)")) return rc;
  bool others = false;
  for (int k = 0; k < MP_OBS_KINDS; ++k) others = others || (k != (int)kind && e->ring[k].base);
  if (others && slots != e->ring_slots)
    return fail(MP_ERR_INVALID, "mp_bind_output_ring: %d slots, but the kinds already bound as rings have %d "
                "(one position for all of them)", (int)slots, e->ring_slots);
  if (!others) { e->ring_slots = slots; e->ring_cursor = 0; }
  e->ring[kind].base = (uint8_t*)base;
  e->ring[kind].stride = slot_stride_bytes;
  for (auto& v : e->ring_plan) v.clear();   // plans belong to the buffers they were timed on
  // between submissions a ring kind points at the slot written last (slot 0 before the first)
  const uint64_t last = e->ring_cursor ? (e->ring_cursor - 1) % (uint64_t)e->ring_slots : 0;
  e->bound[kind] = e->ring[kind].base + last * slot_stride_bytes;
  return MP_OK;
}

int mp_reset(MpEngine* e, const uint64_t* seeds, const uint8_t* mask) {
  if (!e) return fail(MP_ERR_INVALID, "mp_reset: NULL engine");
  e->touched = true;
  HIP_TRY(hipSetDevice(e->device));
  const uint8_t* dmask = nullptr;
  if (mask) {
    HIP_TRY(hipMemcpyAsync(e->d_mask, mask, (size_t)e->N, hipMemcpyHostToDevice, e->stream));
    dmask = e->d_mask;
  }
  if (seeds) {
    HIP_TRY(hipMemcpyAsync(e->d_seeds, seeds, (size_t)e->N * 8, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_set_seeds, dim3((e->N + 255) / 256), dim3(256), 0, e->stream,
                       e->d_state, e->t.world_stride, e->t.grid_pad, e->N,
                       (const uint64_t*)e->d_seeds, dmask);
  }
  if (mask || seeds) HIP_TRY(hipStreamSynchronize(e->stream));  // host buffers are the caller's
  if (!mask && e->h_fault[0] != 0) {
    // a reported pipeline stall stays reported (every synchronising call fails)
    // until ALL worlds are reset: that makes the state whole again
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 0; i < 9; ++i) e->h_fault[i] = 0;
  }
  return submit(e, STEP_MODE_RESET, nullptr, dmask);
}

int mp_step(MpEngine* e, const int32_t* actions_device) {
  if (!e || !actions_device) return fail(MP_ERR_INVALID, "mp_step: NULL argument");
  e->touched = true;
  HIP_TRY(hipSetDevice(e->device));
  return submit(e, STEP_MODE_STEP, actions_device, nullptr);
}

int mp_step_host(MpEngine* e, const int32_t* actions_host) {
  if (!e || !actions_host) return fail(MP_ERR_INVALID, "mp_step_host: NULL argument");
  e->touched = true;
  const size_t NP = (size_t)e->N * e->t.P;
  for (size_t i = 0; i < NP; ++i)
    if (actions_host[i] < 0 || actions_host[i] >= e->t.nact)
      return fail(MP_ERR_INVALID,
                  "mp_step_host: action %d of player %zu in world %zu is outside [0, %d)",
                  actions_host[i], i % e->t.P, i / e->t.P, e->t.nact);
  HIP_TRY(hipSetDevice(e->device));
  // The step kernel reads the actions straight from pinned, device-mapped host
  // memory (28 B per wave, fetched before — and hidden behind — its record load):
  // no copy engine, no stream synchronisation, the host runs ahead of the GPU.
  const int slot = (int)(e->host_steps++ % MpEngine::kHostSlots);
  if (!e->h_actions[slot]) {
    HIP_TRY(hipHostMalloc((void**)&e->h_actions[slot], NP * 4, hipHostMallocMapped));
    HIP_TRY(hipEventCreateWithFlags(&e->h_copied[slot], hipEventDisableTiming));
  } else {
    HIP_TRY(hipEventSynchronize(e->h_copied[slot]));  // the step that read this slot, 4 steps ago
  }
  memcpy(e->h_actions[slot], actions_host, NP * 4);
  int32_t* dev_view = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&dev_view, e->h_actions[slot], 0));
  const int rc = submit(e, STEP_MODE_STEP, dev_view, nullptr);
  HIP_TRY(hipEventRecord(e->h_copied[slot], e->stream));
  return rc;
}

int mp_step_fields(MpEngine* e, const int32_t* fields_device) {
  if (!e || !fields_device) return fail(MP_ERR_INVALID, "mp_step_fields: NULL argument");
  e->touched = true;
  HIP_TRY(hipSetDevice(e->device));
  return submit(e, STEP_MODE_FIELDS, fields_device, nullptr);
}

int mp_step_fields_host(MpEngine* e, const int32_t* fields_host) {
  if (!e || !fields_host) return fail(MP_ERR_INVALID, "mp_step_fields_host: NULL argument");
  e->touched = true;
  const size_t A = (size_t)e->t.nfields, NP = (size_t)e->N * e->t.P;
  for (size_t i = 0; i < NP * A; ++i) {
    const int a = (int)(i % A);
    const int lo = (int)(int8_t)(e->t.field_lo >> (8 * a)), hi = (int)(int8_t)(e->t.field_hi >> (8 * a));
    if (fields_host[i] < lo || fields_host[i] > hi)
      return fail(MP_ERR_INVALID,
                  "mp_step_fields_host: field %d of player %zu in world %zu is %d, outside [%d, %d]",
                  a, (i / A) % e->t.P, i / A / e->t.P, fields_host[i], lo, hi);
  }
  HIP_TRY(hipSetDevice(e->device));
  // (not a hot path: staged through the engine's own device buffer)
  if (!e->d_fields) HIP_TRY(hipMalloc((void**)&e->d_fields, NP * 4 * 4));
  HIP_TRY(hipMemcpyAsync(e->d_fields, fields_host, NP * A * 4, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));   // the host array is the caller's
  return submit(e, STEP_MODE_FIELDS, e->d_fields, nullptr);
}

int mp_observe(MpEngine* e, MpObsKind kind, void* dst) {
  if (!e || !dst) return fail(MP_ERR_INVALID, "mp_observe: NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  const StepOutputs o = e->outputs();
  const void* src = nullptr;
  switch (kind) {
    case MP_OBS_RGB:
      draw(e, (uint8_t*)dst, nullptr);
      HIP_TRY(hipGetLastError());
      return MP_OK;
    case MP_OBS_WORLD_RGB:
      draw(e, nullptr, (uint8_t*)dst);
      HIP_TRY(hipGetLastError());
      return MP_OK;
    case MP_OBS_LAYER:
      launch_layer_view(e->t, e->d_state, (int32_t*)dst, e->N, e->stream);
      HIP_TRY(hipGetLastError());
      return MP_OK;
    case MP_OBS_REWARD: src = o.reward; break;
    case MP_OBS_READY_TO_SHOOT: src = o.ready; break;
    case MP_OBS_AUX0: src = o.aux0; break;
    case MP_OBS_STEP_TYPE: src = o.step_type; break;
    case MP_OBS_DISCOUNT: src = o.discount; break;
    case MP_OBS_COLLECTIVE_REWARD: src = o.collective; break;
    case MP_OBS_POSITION: src = o.position; break;
    case MP_OBS_ORIENTATION: src = o.orientation; break;
    case MP_OBS_EVENTS: src = o.events; break;
    case MP_OBS_AUX1: case MP_OBS_AUX2: case MP_OBS_AUX3: case MP_OBS_AUX4:
      src = o.dbg[kind - MP_OBS_AUX1];
      break;
    case MP_OBS_ZAP_MATRIX: src = o.zap_matrix; break;
    case MP_OBS_INVENTORY: src = o.inventory; break;
    case MP_OBS_INTERACTION_INVENTORIES: src = o.interaction; break;
    case MP_OBS_MATRIX_CUMULANTS: src = o.cumulants; break;
    case MP_OBS_INTERACTION_REWARDS: src = o.interaction_rewards; break;
    default: return fail(MP_ERR_UNSUPPORTED, "mp_observe: unknown observation kind %d", (int)kind);
  }
  if (!src)
    return fail(MP_ERR_UNSUPPORTED,
                "mp_observe: debug observation %d is not produced (bind it before the step, or "
                "create the engine with debug_observations; or the substrate has none)", (int)kind);
  if (src != dst)
    HIP_TRY(hipMemcpyAsync(dst, src, mp_obs_bytes(e, kind), hipMemcpyDeviceToDevice, e->stream));
  return MP_OK;
}

int mp_dump(MpEngine* e, uint8_t* grid, int32_t* avat, int32_t* glob) {
  if (!e || !grid || !avat || !glob) return fail(MP_ERR_INVALID, "mp_dump: NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  if (int rc = sync_and_check(e, "mp_dump")) return rc;
  const DevTables& t = e->t;
  std::vector<uint8_t> host((size_t)e->N * t.world_stride);
  HIP_TRY(hipMemcpy(host.data(), e->d_state, host.size(), hipMemcpyDeviceToHost));
  for (int w = 0; w < e->N; ++w) {
    const uint8_t* rec = host.data() + (size_t)w * t.world_stride;
    const WorldTail* tail = reinterpret_cast<const WorldTail*>(rec + t.grid_pad);
    const size_t render_bytes = (size_t)t.L * t.H * t.W;
    memcpy(grid + (size_t)w * render_bytes, rec, render_bytes);
    for (int p = 0; p < t.P; ++p) {
      int32_t* a = avat + ((size_t)w * t.P + p) * 8;
      a[0] = tail->ax[p]; a[1] = tail->ay[p]; a[2] = tail->aori[p];
      a[3] = tail->aalive[p]; a[4] = tail->ztimer[p]; a[5] = tail->ctimer[p];
      a[6] = tail->frame - tail->achange[p]; a[7] = 0;
    }
    int32_t* g = glob + (size_t)w * 8;
    g[0] = tail->step; g[1] = tail->done; g[2] = tail->frame; g[3] = tail->aux_count;
    g[4] = (int32_t)tail->episode; g[5] = g[6] = g[7] = 0;
    if (e->substrate == MPK_SUBSTRATE_COOP_MINING) {
      // the ores' Lua-side variables, packed as oracle/coop_mining.c:coop_dump packs them:
      // the sum of the live countdowns, a position-weighted sum of the miner sets
      const CoopTables& c = e->cm;
      const int32_t* cells = table<int32_t>(e->pack.data(), "ore_cells");
      const uint8_t* M = rec + (size_t)c.plane_m * t.H * t.W;
      const uint8_t* C = rec + (size_t)c.plane_c * t.H * t.W;
      uint32_t cd = 0, ms = 0;
      for (int i = 0; i < c.n_ore; ++i) {
        cd += C[cells[i]];
        ms += (uint32_t)M[cells[i]] * (uint32_t)(i + 1);
      }
      g[5] = (int32_t)cd; g[6] = (int32_t)(ms & 0x7fffffffu);
    }
    if (e->substrate == MPK_SUBSTRATE_COLLABORATIVE_COOKING) {
      // the pots' cooking times, summed as oracle/collaborative_cooking.c:cook_dump sums them
      const CookTables& c = e->cc;
      const int32_t* pots = table<int32_t>(e->pack.data(), "cc_pot_cells");
      const uint8_t* T = rec + (size_t)c.plane_t * t.H * t.W;
      uint32_t times = 0;
      for (int k = 0; k < c.n_pot; ++k) times += (uint32_t)(T[pots[k]] & 31) * (uint32_t)(k + 1);
      g[3] = 0; g[5] = (int32_t)times;
    }
    if (e->substrate == MPK_SUBSTRATE_GIFT_REFINEMENTS) {
      // the inventories, packed as oracle/gift_refinements.c:gift_dump packs them
      for (int p = 0; p < t.P; ++p)
        avat[((size_t)w * t.P + p) * 8 + 7] = tail->flag0[p] | (tail->flag1[p] << 4) | (tail->level[p] << 8);
    }
    if (e->substrate == MPK_SUBSTRATE_THE_MATRIX) {
      // extra parity fields, same packing as oracle/the_matrix.c:matrix_dump
      const MatrixTables& c = e->mx;
      for (int p = 0; p < t.P; ++p) {
        int32_t* a = avat + ((size_t)w * t.P + p) * 8;
        const int f1 = tail->flag1[p];
        a[5] = tail->level[p] | ((f1 & 7) << 8) | (((f1 >> 3) & 1) << 12) |
               ((tail->aflags[p] & 1) << 13) | (tail->freeze[p] << 16);
        const int m = tail->flag0[p];
        a[7] = m ? (1 | (tail->ctimer[p] << 1) | (tail->nozap[p] << 9) |
                    ((int)((c.s_mark_packed >> (8 * (m - 1))) & 255ull) << 17))
                 : 0;
      }
      const uint8_t* A = rec + (size_t)c.plane_a * t.H * t.W;
      for (int cell = 0; cell < t.H * t.W; ++cell)
        if ((A[cell] >> 4) & 1) g[5] += A[cell] & 3;
    }
    if (e->substrate == MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS) {
      // extra parity fields, same packing as oracle/externality_mushrooms.c:mushroom_dump
      const MushroomTables& c = e->em;
      for (int p = 0; p < t.P; ++p) {
        avat[((size_t)w * t.P + p) * 8 + 5] = 0;   // (ctimer holds the marking's x here, not a timer)
        avat[((size_t)w * t.P + p) * 8 + 7] =
            tail->level[p] | (tail->freeze[p] << 4) | (tail->removal[p] << 12) |
            (tail->nozap[p] << 16) | ((tail->aflags[p] & 1) << 24) |
            (((tail->aflags[p] >> 1) & 1) << 25);
      }
      const int32_t* cells = table<int32_t>(e->pack.data(), "mushroom_cells");
      const uint8_t* S = rec + (size_t)c.live_layer * t.H * t.W;
      const uint8_t* A = rec + (size_t)c.plane_age * t.H * t.W;
      int live = 0, ages = 0;
      for (int i = 0; i < c.n_site; ++i)
        if (S[cells[i]] != 0) { live++; ages += (S[cells[i]] - c.s_type0 + 1) * A[cells[i]]; }
      g[3] = live; g[5] = ages; g[6] = tail->aux_count - c.n_live_init + 1000; g[7] = tail->aux_count;
    }
    if (e->substrate == MPK_SUBSTRATE_TERRITORY) {
      // extra parity fields, same packing as oracle/territory.c:territory_dump
      for (int p = 0; p < t.P; ++p)
        avat[((size_t)w * t.P + p) * 8 + 7] =
            tail->level[p] | (tail->freeze[p] << 4) | (tail->removal[p] << 12) |
            (tail->nozap[p] << 16) | ((tail->aflags[p] & 1) << 24) |
            (((tail->aflags[p] >> 1) & 1) << 25);
      const uint8_t* A = rec + (size_t)e->tr.plane_a * t.H * t.W;
      std::vector<int32_t> cells((size_t)e->tr.n_res);
      memcpy(cells.data(), table<int32_t>(e->pack.data(), "resource_cells"),
             cells.size() * sizeof(int32_t));
      for (int32_t cell : cells) {
        g[5] += A[cell] & 3; g[6] += (A[cell] >> 2) & 1; g[7] += A[cell] >> 3;
      }
    }
  }
  return MP_OK;
}

uint64_t mp_snapshot_bytes(const MpEngine* e) {
  return e ? (uint64_t)e->N * e->t.world_stride : 0;
}

int mp_snapshot(MpEngine* e, void* buf, uint64_t bytes) {
  if (!e || !buf || bytes != mp_snapshot_bytes(e))
    return fail(MP_ERR_INVALID, "mp_snapshot: bad buffer");
  HIP_TRY(hipSetDevice(e->device));
  if (int rc = sync_and_check(e, "mp_snapshot")) return rc;
  HIP_TRY(hipMemcpy(buf, e->d_state, bytes, hipMemcpyDeviceToHost));
  return MP_OK;
}

int mp_restore(MpEngine* e, const void* buf, uint64_t bytes) {
  if (!e || !buf || bytes != mp_snapshot_bytes(e))
    return fail(MP_ERR_INVALID, "mp_restore: bad buffer");
  HIP_TRY(hipSetDevice(e->device));
  e->touched = true;
  if (int rc = sync_and_check(e, "mp_restore")) return rc;
  HIP_TRY(hipMemcpy(e->d_state, buf, bytes, hipMemcpyHostToDevice));
  return MP_OK;
}

int mp_counters(MpEngine* e, uint64_t out[MP_CTR_COUNT]) {
  if (!e || !out) return fail(MP_ERR_INVALID, "mp_counters: NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipMemsetAsync(e->d_ctr, 0, MP_CTR_COUNT * 8, e->stream));
  hipLaunchKernelGGL(k_sum_counters, dim3(256), dim3(256), 0, e->stream, e->d_state,
                     e->t.world_stride, e->t.grid_pad, e->N, e->d_ctr);
  HIP_TRY(hipGetLastError());
  unsigned long long host[MP_CTR_COUNT];
  HIP_TRY(hipMemcpyAsync(host, e->d_ctr, sizeof(host), hipMemcpyDeviceToHost, e->stream));
  if (int rc = sync_and_check(e, "mp_counters")) return rc;
  for (int k = 0; k < MP_CTR_COUNT; ++k) out[k] = host[k];
  return MP_OK;
}

#if defined(MP_FRAME_TIMELINE) || defined(MP_FRAME_ENDS)
// developer build: the frame kernel's event log (frame.hip: FRAME_STAGE) / its per-workgroup stamps
__attribute__((visibility("default")))   // (not in the header: these builds only)
int mp_debug_timeline(MpEngine* e, uint32_t* out, int nwords) {
  if (!e || !out || nwords > kFaultWords - 64) return MP_ERR_INVALID;
#if defined(MP_FRAME_ENDS)
  if (nwords > 2 * 1024) return MP_ERR_INVALID;
  (void)hipStreamSynchronize(e->stream);
  (void)hipMemcpy(out, e->d_claim + 2, (size_t)nwords * 4, hipMemcpyDeviceToHost);
  (void)hipMemset(e->d_claim + 2, 0, 2 * 1024 * 4);
  return MP_OK;
#endif
  for (int i = 0; i < nwords; ++i) out[i] = ((const volatile uint32_t*)e->h_fault)[64 + i];
  memset((void*)(e->h_fault + 64), 0, (size_t)(kFaultWords - 64) * 4);
  return MP_OK;
}
#endif

// ---- memory for a bound view (include/mp_engine.h: mp_alloc_output)
}  // extern "C"
namespace {
struct MappedView { size_t bytes; size_t chunk; std::vector<hipMemGenericAllocationHandle_t> handles; };
std::map<void*, MappedView> g_mapped;   // views made of mapped chunks (plain ones are not listed)
std::mutex g_mapped_lock;
// Address space retired by released mapped views (free_output keeps their ranges reserved), and
// the bound beyond which nothing more is mapped: a process that places views for ever (a sweep
// that creates engine after engine) gets a clear error instead of an address space that silently
// fills up.  16 TiB: ~25,000 placed clean_up views.
int64_t g_retired_va = 0;
int64_t g_retired_va_limit = (int64_t)16 << 40;

void retired_va(int64_t* bytes, int64_t* limit) {
  std::lock_guard<std::mutex> g(g_mapped_lock);
  if (bytes) *bytes = g_retired_va;
  if (limit) *limit = g_retired_va_limit;
}

// is `p` inside a view this library mapped?  (hipPointerGetAttributes does not know them)
bool in_mapped_view(const void* p) {
  std::lock_guard<std::mutex> g(g_mapped_lock);
  auto it = g_mapped.upper_bound(const_cast<void*>(p));
  if (it == g_mapped.begin()) return false;
  --it;
  return (const char*)p < (const char*)it->first + it->second.bytes;
}

// undoes a partly built mapping (best effort) and reports `rc`
int mapped_failed(void* base, MappedView& v, size_t mapped, hipError_t rc, const char* what) {
  if (base) {
    if (mapped) (void)hipMemUnmap(base, mapped * v.chunk);
    (void)hipMemAddressFree(base, v.bytes);
  }
  for (auto h : v.handles) (void)hipMemRelease(h);
  (void)hipGetLastError();
  return fail(MP_ERR_HIP, "mp_alloc_output: %s failed: %s", what, hipGetErrorString(rc));
}
}  // namespace
extern "C" {

// One virtual range mapped onto separately created physical chunks: the view the frame launch
// writes evenly (a view whose physical pages lie next to each other — a contiguous extent,
// large pieces of a plain allocation — is written 25 - 45 % slower: profiles/r05_alloc_method.md;
// scattering the chunks FURTHER, pools and shuffles, added nothing there and is gone).
static int alloc_mapped(int device, uint64_t bytes, uint64_t chunk_bytes, void** out) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, chunk_bytes >= (2u << 20)
                                                           ? hipMemAllocationGranularityRecommended
                                                           : hipMemAllocationGranularityMinimum));
  if (gran == 0) gran = 4096;
  MappedView v;
  v.chunk = ((size_t)chunk_bytes + gran - 1) / gran * gran;
  const size_t n = ((size_t)bytes + v.chunk - 1) / v.chunk;
  if (n > (1u << 20)) return fail(MP_ERR_INVALID, "mp_alloc_output: %zu chunks", n);
  v.bytes = n * v.chunk;
  {
    int64_t retired = 0, limit = 0;
    retired_va(&retired, &limit);
    if (retired + (int64_t)v.bytes > limit)
      return fail(MP_ERR_HIP, "mp_alloc_output: this process has retired %lld bytes of address space with "
                  "released mapped views (their ranges are never reused: stale translations); mapping %zu "
                  "more would pass the bound of %lld (mp_set_retired_va_limit) — reuse views instead of "
                  "placing new ones", (long long)retired, v.bytes, (long long)limit);
  }
  void* base = nullptr;
  hipError_t rc = hipMemAddressReserve(&base, v.bytes, v.chunk < (2u << 20) ? (2u << 20) : v.chunk, nullptr, 0);
  if (rc != hipSuccess) return mapped_failed(nullptr, v, 0, rc, "hipMemAddressReserve");
  v.handles.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    hipMemGenericAllocationHandle_t h;
    rc = hipMemCreate(&h, v.chunk, &prop, 0);
    if (rc != hipSuccess) return mapped_failed(base, v, 0, rc, "hipMemCreate");
    v.handles.push_back(h);
  }
  for (size_t i = 0; i < n; ++i) {
    rc = hipMemMap((char*)base + i * v.chunk, v.chunk, 0, v.handles[i], 0);
    if (rc != hipSuccess) return mapped_failed(base, v, i, rc, "hipMemMap");
  }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  rc = hipMemSetAccess(base, v.bytes, &acc, 1);
  if (rc != hipSuccess) return mapped_failed(base, v, n, rc, "hipMemSetAccess");
  {
    std::lock_guard<std::mutex> g(g_mapped_lock);
    g_mapped[base] = v;
  }
  *out = base;
  return MP_OK;
}

int mp_alloc_output(int device, uint64_t bytes, uint64_t chunk_bytes, void** out) {
  if (!out || bytes == 0) return fail(MP_ERR_INVALID, "mp_alloc_output: bad argument");
  *out = nullptr;
  HIP_TRY(hipSetDevice(device));
  if (chunk_bytes == 0) {
    const hipError_t rc = hipMalloc(out, (size_t)bytes);
    if (rc != hipSuccess) {
      (void)hipGetLastError();
      *out = nullptr;
      return fail(MP_ERR_HIP, "mp_alloc_output: hipMalloc of %llu bytes failed: %s",
                  (unsigned long long)bytes, hipGetErrorString(rc));
    }
    return MP_OK;
  }
  return alloc_mapped(device, bytes, chunk_bytes, out);
}

// The physical chunks go back to the driver; the VIRTUAL range stays reserved and is
// never handed out again (`keep_va`, always true in this library).  Measured on this
// stack (ROCm 7.2, gfx950): a range released with hipMemAddressFree is reused by the next
// hipMemAddressReserve, and kernels then write through translations of the OLD mapping —
// a second placed view came back with 267 - 1030 of 1030 worlds stale, no error anywhere
// (tools/gpu_r04_dbg_place.py).  Address space is not scarce (a placement retires
// ~12 x the view's size of it); correctness is.
static int free_output(int device, void* ptr, bool keep_va) {
  if (!ptr) return MP_OK;
  HIP_TRY(hipSetDevice(device));
  MappedView v;
  bool mapped = false;
  {
    std::lock_guard<std::mutex> g(g_mapped_lock);
    auto it = g_mapped.find(ptr);
    if (it != g_mapped.end()) { v = it->second; g_mapped.erase(it); mapped = true; }
  }
  if (!mapped) {
    HIP_TRY(hipFree(ptr));   // (waits for the device's work on the buffer)
    return MP_OK;
  }
  // best effort: whatever fails, the rest is still released
  hipError_t first = hipDeviceSynchronize(), rc = hipMemUnmap(ptr, v.bytes);
  if (first == hipSuccess) first = rc;
  for (auto h : v.handles) {
    rc = hipMemRelease(h);
    if (first == hipSuccess) first = rc;
  }
  if (!keep_va) {
    rc = hipMemAddressFree(ptr, v.bytes);
    if (first == hipSuccess) first = rc;
  } else {
    std::lock_guard<std::mutex> g(g_mapped_lock);
    g_retired_va += (int64_t)v.bytes;
  }
  if (first != hipSuccess) {
    (void)hipGetLastError();
    return fail(MP_ERR_HIP, "mp_free_output: %s", hipGetErrorString(first));
  }
  return MP_OK;
}

int mp_free_output(int device, void* ptr) { return free_output(device, ptr, true); }

// torch.cuda.memory.CUDAPluggableAllocator entry points (meltingpot_amd/memory.py): tensors a
// caller allocates inside `memory.mapped_allocations()` — a learner's own rollout buffers —
// come from scattered 2 MB chunks like the engine's own views (32 MB and up; below that an
// ordinary hipMalloc).  NULL on failure, as the allocator interface expects.
void* mp_torch_alloc(ssize_t size, int device, void* stream) {
  (void)stream;
  void* p = nullptr;
  if (size <= 0) return nullptr;
  if (mp_alloc_output(device, (uint64_t)size, size >= (ssize_t)(32 << 20) ? (2u << 20) : 0, &p) != MP_OK)
    return nullptr;
  return p;
}

void mp_torch_free(void* ptr, ssize_t size, int device, void* stream) {
  (void)size; (void)stream;
  (void)free_output(device, ptr, true);
}

int mp_set_retired_va_limit(int64_t bytes) {
  if (bytes < 0) return fail(MP_ERR_INVALID, "mp_set_retired_va_limit: %lld", (long long)bytes);
  std::lock_guard<std::mutex> g(g_mapped_lock);
  g_retired_va_limit = bytes;
  return MP_OK;
}

namespace {

// The launches mp_tune times, back to back (one pair of events around `reps` of them,
// two more in front: the device stays busy, the clocks where a training loop has them):
// dry — a reset whose mask names no world: nothing is stepped, no record written back,
// every bound view drawn exactly as a step draws it — or, on an engine nothing has been
// done with yet, REAL steps (uniformly random actions) behind a device-side copy of the state.
struct Events {   // RAII: a pair of timing events
  hipEvent_t a = nullptr, b = nullptr;
  int create() {
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    return MP_OK;
  }
  ~Events() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};

int timed_launches_us(MpEngine* e, bool real, int reps, double* us) {
  Events ev;
  if (int rc = ev.create()) return rc;
  int rc = MP_OK;
  auto one = [&]() {
    return real ? submit(e, STEP_MODE_STEP, e->d_actions, nullptr)
                : submit(e, STEP_MODE_RESET, nullptr, e->d_mask);
  };
  for (int r = 0; r < 2 && rc == MP_OK; ++r) rc = one();
  HIP_TRY(hipEventRecord(ev.a, e->stream));
  for (int r = 0; r < reps && rc == MP_OK; ++r) rc = one();
  HIP_TRY(hipEventRecord(ev.b, e->stream));
  if (hipEventSynchronize(ev.b) != hipSuccess) {
    (void)hipGetLastError();
    if (rc == MP_OK) rc = fail(MP_ERR_HIP, "mp_tune: a probe launch failed");
  }
  float ms = 0;
  if (rc == MP_OK) HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
  *us = (double)ms * 1e3 / (reps > 0 ? reps : 1);
  return rc;
}

// What a probe that really steps must put back, whatever happens in between (RAII: every
// exit path of mp_tune — and of mp_place_output, which calls it — leaves the engine the engine
// it was): the records, the counters and the engine's own scalar outputs are copied aside and
// copied back; the CALLER's scalar outputs are unbound for the duration (the probe's steps
// write the engine's own buffers instead), so that no memory of the caller's but the pixel
// views being timed is touched.
struct ProbeState {
  MpEngine* e;
  uint8_t* saved = nullptr;
  bool copied = false;
  bool held = false, was_held = false;
  void* rebind[MP_OBS_KINDS] = {};
  size_t parts[4] = {};
  explicit ProbeState(MpEngine* eng) : e(eng) {
    parts[0] = (size_t)e->N * e->t.world_stride;
    parts[1] = MP_CTR_COUNT * 8;
    parts[2] = e->scalars_bytes;
    parts[3] = e->d_debug ? e->debug_bytes : 0;
  }
  uint8_t* part(int k) const {
    return k == 0 ? e->d_state : k == 1 ? (uint8_t*)e->d_ctr : k == 2 ? e->d_scalars : e->d_debug;
  }
  void hold_ring() { was_held = e->ring_hold; e->ring_hold = true; held = true; }
  // MP_OK with copied == false: no room for the copy — the probe runs dry
  int save() {
    const size_t total = parts[0] + parts[1] + parts[2] + parts[3];
    if (hipMalloc((void**)&saved, total) != hipSuccess) {
      (void)hipGetLastError();
      saved = nullptr;
      return MP_OK;
    }
    size_t off = 0;
    for (int k = 0; k < 4; ++k) {
      if (parts[k]) {
        const hipError_t rc = hipMemcpyAsync(saved + off, part(k), parts[k], hipMemcpyDeviceToDevice, e->stream);
        if (rc != hipSuccess) {
          (void)hipGetLastError();
          (void)hipStreamSynchronize(e->stream);
          (void)hipFree(saved);
          saved = nullptr;
          return fail(MP_ERR_HIP, "mp_tune: saving the engine's state failed: %s", hipGetErrorString(rc));
        }
      }
      off += parts[k];
    }
    copied = true;
    for (int k = 0; k < MP_OBS_KINDS; ++k)
      if (k != MP_OBS_RGB && k != MP_OBS_WORLD_RGB) { rebind[k] = e->bound[k]; e->bound[k] = nullptr; }
    return MP_OK;
  }
  int restore() {
    int rc = MP_OK;
    if (copied) {
      size_t off = 0;
      hipError_t first = hipSuccess;
      for (int k = 0; k < 4; ++k) {
        if (parts[k]) {
          const hipError_t r = hipMemcpyAsync(part(k), saved + off, parts[k], hipMemcpyDeviceToDevice, e->stream);
          if (first == hipSuccess) first = r;
        }
        off += parts[k];
      }
      const hipError_t r = hipStreamSynchronize(e->stream);
      if (first == hipSuccess) first = r;
      for (int k = 0; k < MP_OBS_KINDS; ++k)
        if (k != MP_OBS_RGB && k != MP_OBS_WORLD_RGB) e->bound[k] = rebind[k];
      copied = false;
      if (first != hipSuccess) {
        (void)hipGetLastError();
        rc = fail(MP_ERR_HIP, "mp_tune: putting the engine's state back failed: %s", hipGetErrorString(first));
      }
    }
    if (saved) { (void)hipFree(saved); saved = nullptr; }
    if (held) {
      e->ring_hold = was_held;
      held = false;
      if (e->ring_slots > 0 && !e->ring_hold) {   // ring kinds point at the slot written last again
        const uint64_t last = e->ring_cursor ? (e->ring_cursor - 1) % (uint64_t)e->ring_slots : 0;
        e->point_ring((int)last);
      }
    }
    return rc;
  }
  ~ProbeState() { (void)restore(); }
};

}  // namespace

static int tune_impl(MpEngine* e, double* us_per_launch, bool* stepped, bool quick = false);

int mp_tune(MpEngine* e, double* us_per_launch) { return tune_impl(e, us_per_launch, nullptr); }

// (`stepped`: whether the probe really stepped — false when the engine is in use or there
// was no room for the copy of its state; `quick`: what mp_place_output asks of every candidate
// buffer — the stock plan and the team order, one group of launches each, no stages: buffers
// differ by 10 - 25 %, the winner gets the whole search afterwards)
static int tune_impl(MpEngine* e, double* us_per_launch, bool* stepped, bool quick) {
  if (!e) return fail(MP_ERR_INVALID, "mp_tune: NULL engine");
  HIP_TRY(hipSetDevice(e->device));
  if (stepped) *stepped = false;
  if (us_per_launch) *us_per_launch = 0.0;
  // a ring: every slot is its own buffer (its own physical pages), the plan follows each
  const int slots = e->ring_slots > 0 && e->ring_has_pixels() ? e->ring_slots : 1;
  uint8_t* rgb = (uint8_t*)e->bound[MP_OBS_RGB];
  uint8_t* wrgb = (uint8_t*)e->bound[MP_OBS_WORLD_RGB];
  if ((!rgb && !wrgb) || !e->fuse(rgb == nullptr)) return MP_OK;
  const int views = rgb && wrgb ? 2 : wrgb ? 1 : 0;
  // (everything in flight finishes first: a tune between two steps sees whole records)
  if (int rc = sync_and_check(e, "mp_tune")) return rc;
  FramePlan& plan = e->plan[1][views];
  const FramePlan before = plan;
  const FramePlan stock = plan_frame(e->t, e->sub, e->N, true, views, e->num_cus, nullptr);
  // the candidates: the stock plan; the same ring cut into single worlds; that with
  // half of every workgroup's share pooled; the stock plan with sc1 stores.  (Same
  // number of LDS record slots: the composite cache was sized for the stock plan.)
  std::vector<FramePlan> cand;
  cand.push_back(e->has_dev ? plan : stock);
  if (!e->has_dev) {
    const int lds_slots = stock.NB * stock.B;
    for (int pct : {100, 50}) {
      if (quick) break;
      MpDevOptions d = {};
      d.struct_size = sizeof d;
      d.max_composites = -1;
      d.batch_worlds = 1;
      d.ring_batches = lds_slots;
      d.static_pct = pct;
      const FramePlan p = plan_frame(e->t, e->sub, e->N, true, views, e->num_cus, &d);
      if (frame_lds_bytes(e->t, p) <= frame_lds_bytes(e->t, stock) &&
          (p.B != stock.B || p.NB != stock.NB || p.pool != stock.pool))
        cand.push_back(p);
    }
    // ... the single-world ring dealt to XCD teams (round 6: each XCD writes one compact front; since
    // the launch is its own store loop that is 118 -> 109 us for WORLD.RGB on a view the memory side
    // serves unevenly, 94 -> 89 on one it serves evenly, and with a pause on top 100: profiles/r06_resolve.md)
    {
      MpDevOptions d = {};
      d.struct_size = sizeof d;
      d.max_composites = -1;
      d.batch_worlds = 1;
      d.ring_batches = lds_slots;
      d.team = 1;
      const FramePlan p = plan_frame(e->t, e->sub, e->N, true, views, e->num_cus, &d);
      if (p.team && frame_lds_bytes(e->t, p) <= frame_lds_bytes(e->t, stock)) {
        cand.push_back(p);
        if (!quick && views != 1 && stock.feeders >= 4) {   // ... and with half the feeders (see below)
          d.feeders = stock.feeders / 2;
          const FramePlan h = plan_frame(e->t, e->sub, e->N, true, views, e->num_cus, &d);
          if (h.team && h.feeders != p.feeders && frame_lds_bytes(e->t, h) <= frame_lds_bytes(e->t, stock))
            cand.push_back(h);
        }
      }
    }
    // ... and the stock ring with sc1 pixel stores: 13 % faster for commons_harvest on
    // the buffers the memory side serves unevenly (341 -> 297 us), slower everywhere
    // else (profiles/r04_plans.md)
    // (per-agent views only: for WORLD.RGB it is slower on every buffer measured, by more
    // than a probe of NOOP steps resolves)
    if (views != 1 && !quick) {
      FramePlan q = stock;
      q.store_sc1 = 1;
      cand.push_back(q);
    }
    // ... and half the feeders (6 -> 3: three more drawing waves).  Where the memory side
    // serves a buffer evenly the per-agent drawing is issue-bound and the extra waves are
    // worth 8 - 13 % (clean_up, both views: 244 -> 205 - 213 us); where it does not, the
    // thirteenth wave starves and the launch is 4 % SLOWER (profiles/r04_head.md)
    if (views != 1 && stock.feeders >= 4 && !quick) {
      MpDevOptions d = {};
      d.struct_size = sizeof d;
      d.max_composites = -1;
      d.feeders = stock.feeders / 2;
      const FramePlan p = plan_frame(e->t, e->sub, e->N, true, views, e->num_cus, &d);
      if (frame_lds_bytes(e->t, p) <= frame_lds_bytes(e->t, stock) && p.feeders != stock.feeders)
        cand.push_back(p);
    }
  }
  if (cand.size() == 1 && !us_per_launch) return MP_OK;
  // An engine nothing has been done with yet (the usual moment to bind) is really
  // stepped: all worlds reset, uniformly random actions, behind a device-side copy of the records,
  // the counters and the engine's scalar outputs — what a plan costs when it steps is what is wanted, and a dry
  // launch ranks plans a few per cent apart wrongly (measured: the single-world ring
  // 96.5 us dry, 106.7 stepping, against 96.7 / 103.0 for the stock ring:
  // profiles/r04_plans.md).  An engine in use is timed dry, and a plan must then beat
  // the stock one by 6 % to replace it.
  ProbeState probe(e);   // (its destructor puts everything back on every path out of here)
  probe.hold_ring();
  if (!e->touched)
    if (int rc = probe.save()) return rc;
  const bool stepping = probe.copied;
  if (stepped) *stepped = stepping;
  if (e->ring_slots > 0) e->point_ring(0, stepping);
  int rc = MP_OK;
  if (stepping) {
    const int n = e->N * e->t.P;
    hipLaunchKernelGGL(k_probe_actions, dim3((n + 255) / 256), dim3(256), 0, e->stream,
                       e->d_actions, n, e->t.nact, 0x5eedu);
    rc = submit(e, STEP_MODE_RESET, nullptr, nullptr);
  } else {
    const hipError_t r = hipMemsetAsync(e->d_mask, 0, (size_t)e->N, e->stream);
    if (r != hipSuccess) { (void)hipGetLastError(); rc = fail(MP_ERR_HIP, "mp_tune: %s", hipGetErrorString(r)); }
  }
  // A device that has idled for a few ms runs its next ~150 launches 5 - 20 % slower
  // (clock ramp, profiles/r03_clock_ramp.md) — which would be charged to whichever plan
  // is timed first.  The first candidate runs untimed, in groups of eight, until five
  // groups in a row are within 1 % of the fastest group so far and none of them has
  // improved on it by 0.5 % (at most 40 groups, 32 ms): the ramp is not monotonic — 120 110
  // 117 118 116 114 112 111 110 110 108 107 107 107 us by tens of launches after 1 s of
  // idling — and two groups agreeing, round 3's rule, can be a plateau half way up.  The
  // clocks are then where a training loop, which never lets the device idle, has them —
  // for the candidates' timings and for whatever the caller launches next.
  plan = cand[0];
  {
    double best = 1e30;
    int steady = 0;
    for (int g = 0; g < 40 && rc == MP_OK && steady < 5; ++g) {
      double us = 0.0;
      rc = timed_launches_us(e, stepping, 6, &us);   // (2 + 6 launches)
      if (us < 0.995 * best) { best = us; steady = 0; }
      else if (us <= 1.01 * best) { ++steady; if (us < best) best = us; }
      else steady = 0;
    }
  }
  std::vector<FramePlan> kept((size_t)slots, cand[0]);
  double sum_us = 0;
  for (int sl = 0; sl < slots && rc == MP_OK; ++sl) {
    if (e->ring_slots > 0) e->point_ring(sl, stepping);
    double best_us = 1e30, stock_us = 0;
    int best = 0;
    std::vector<double> first_us(cand.size(), 1e30);
    for (size_t i = 0; i < cand.size() && rc == MP_OK; ++i) {
      plan = cand[i];
      rc = timed_launches_us(e, stepping, 6, &first_us[i]);
    }
    // A second look at whatever came within 6 % of the fastest, three times as long (round 6: with eight
    // candidates a few per cent apart — the team order is worth 3 - 5 % on an even buffer — one group of
    // six launches picked differently from run to run): the stock plan always, the others by their first
    // timing.  (A plan replaces the stock one only by a margin: the probe's steps are the first of an
    // episode, or no steps at all — 3 % stepping, 6 % dry.)
    {
      double fastest = 1e30;
      for (double us : first_us) fastest = std::min(fastest, us);
      for (size_t i = 0; i < cand.size() && rc == MP_OK; ++i) {
        if (i != 0 && first_us[i] > 1.06 * fastest) continue;
        plan = cand[i];
        double us = first_us[i];
        if (!quick) rc = timed_launches_us(e, stepping, 18, &us);
        if (i == 0) stock_us = us;
        if (rc == MP_OK && (i == 0 || us < std::min(best_us, (stepping ? 0.97 : 0.94) * stock_us))) {
          best_us = us; best = (int)i;
        }
      }
    }
    // ... and the pause of a renderer wave between two passes (FramePlan::pace, round 6).  Since the
    // renderers' resolve is two LDS round trips a pass instead of eighteen, the launch is its own store loop plus the head
    // on every buffer — 90 - 95 us for WORLD.RGB where the memory side takes the view's pages evenly, and
    // 113 - 119 where it does not: there a launch that writes FASTER finishes LATER (the old resolve's
    // 105 us on such a buffer were its pace), and a pause of two or three units gives the 104 back
    // (profiles/r06_resolve.md).  Searched on the plan just picked; the same margin as between plans.
    FramePlan chosen = cand[(size_t)best];
    // ... the feeders' wave priority once their first world is out (FramePlan::late_prio; round 6).  Under
    // the new resolve the renderers issue instructions where the old one waited on LDS, and a feeder at
    // their priority steps more slowly beside them: where the steps are the long pole (sixteen worlds
    // behind four feeders: externality_mushrooms, coop_mining, gift_refinements) priority 1 is 4 - 5 %
    // on every buffer, for commons_harvest 3 - 6 %; for clean_up's per-agent view it costs 4 % on an even
    // buffer (profiles/r06_resolve.md section 8) — so it is timed, on the plan just picked.
    // (only a probe that really steps can see it: in a dry launch the feeders load records and nothing else)
    if (rc == MP_OK && !e->has_dev && stepping && chosen.late_prio == 0 && !quick) {
      FramePlan q = chosen;
      q.late_prio = 1;
      plan = q;
      double us = 0;
      rc = timed_launches_us(e, stepping, 12, &us);
      if (rc == MP_OK && us < (stepping ? 0.97 : 0.94) * best_us) { best_us = us; chosen = q; }
    }
    if (rc == MP_OK && !e->has_dev && !quick) {
      // (all five: the response is not monotonic — clean_up's per-agent view on an uneven buffer 169.8 /
      // 169.7 / 166.2 / 154.7 / 166.0 us at 0 / 1 / 2 / 4 / 6 units, commons_harvest 345 / 341 / 329 / 310 / 337)
      const FramePlan base = chosen;
      const double unpaced_us = best_us;
      for (int pc : {1, 2, 3, 4, 6}) {
        if (rc != MP_OK) break;
        FramePlan q = base;
        q.pace = pc;
        plan = q;
        double us = 0;
        rc = timed_launches_us(e, stepping, 12, &us);
        // (3 % dry or stepping: what a pause changes is the renderers against the memory side, which a
        // dry launch has whole; the 6 % of a dry probe is for plans that move the FEEDERS' work)
        if (rc == MP_OK && us < std::min(best_us, 0.97 * unpaced_us)) { best_us = us; chosen = q; }
      }
    }
    if (rc == MP_OK) { kept[(size_t)sl] = chosen; sum_us += best_us; }
  }
  plan = rc == MP_OK ? kept[0] : before;
  if (rc == MP_OK && e->ring_slots > 0 && e->ring_has_pixels()) e->ring_plan[views] = kept;
  const int rc2 = probe.restore();   // ... and the engine is the engine it was
  if (rc != MP_OK) return rc;
  if (rc2 != MP_OK) return rc2;
  if (us_per_launch) *us_per_launch = sum_us / slots;
  return sync_and_check(e, "mp_tune");
}

int mp_place_output(MpEngine* e, MpObsKind kind, int32_t candidates, uint64_t max_bytes,
                    void** device_ptr, MpPlacement* report) {
  if (!e || !device_ptr) return fail(MP_ERR_INVALID, "mp_place_output: NULL argument");
  *device_ptr = nullptr;
  if (report) memset(report, 0, sizeof *report);
  if (kind != MP_OBS_RGB && kind != MP_OBS_WORLD_RGB)
    return fail(MP_ERR_INVALID, "mp_place_output: kind %d is not a pixel view", (int)kind);
  if (e->ring[kind].base)
    return fail(MP_ERR_INVALID, "mp_place_output: kind %d is bound as a ring; unbind it first", (int)kind);
  if (candidates < 1) candidates = 1;
  if (candidates > 32) candidates = 32;
  HIP_TRY(hipSetDevice(e->device));
  const auto t0 = std::chrono::steady_clock::now();
  const uint64_t bytes = mp_obs_bytes(e, kind);
  if (bytes == 0)
    return fail(MP_ERR_UNSUPPORTED, "mp_place_output: this substrate has no observation %d", (int)kind);
  if (max_bytes == 0) {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    max_bytes = free_b / 4;
  }
  // candidates alive at a time: released chunks come straight back from the driver's
  // pool, so a round's buffers are held together to be different placements
  uint64_t alive = max_bytes / bytes;
  if (alive < 1)
    return fail(MP_ERR_INVALID, "mp_place_output: max_bytes %llu holds no view of %llu bytes",
                (unsigned long long)max_bytes, (unsigned long long)bytes);
  if (alive > 12) alive = 12;   // a round; another one only if no candidate of it stands out
  if (alive > (uint64_t)candidates) alive = (uint64_t)candidates;
  if (alive == 1) candidates = 1;   // nothing can be compared inside the caller's bound
  // RAII: whatever path leaves this function, every buffer but the one handed out is released
  // and the kind is bound to what it was bound to (or to the winner)
  struct Round {
    MpEngine* e; MpObsKind kind; void* previous; void* keep = nullptr;
    std::vector<void*> bufs;
    bool done = false;
    ~Round() {
      for (void* p : bufs)
        if (p != keep) (void)free_output(e->device, p, true);
      if (!done) {
        if (keep) (void)free_output(e->device, keep, true);
        e->bound[kind] = previous;
      }
    }
    void release_losers() {
      for (void* p : bufs)
        if (p != keep) (void)free_output(e->device, p, true);
      bufs.clear();
    }
  } round{e, kind, e->bound[kind]};
  MpPlacement rep = {};
  rep.requested = candidates;
  double best_us = 1e30;
  int rc = MP_OK;
  bool first = true, exhausted = false;
  while (rep.candidates < candidates && rc == MP_OK && !exhausted) {
    // a round: as many fresh buffers as fit next to the best one so far — the FIRST round
    // eight at most: where a box shows no spread between views mapped from scattered 2 MB
    // chunks (profiles/r05_alloc_method.md), eight that agree to 3 % settle it in half a
    // second; four were too few — on a box where every second candidate is the slow kind
    // (this round's per-agent view: 124 - 128 against 145 - 157 us) one first round in
    // sixteen is all slow, agrees with itself, and keeps a view 25 % slower than the next
    uint64_t room = alive - (round.keep && alive > 1 ? 1 : 0);
    if (rep.candidates == 0 && room > 8) room = 8;
    for (uint64_t i = 0; i < room && rep.candidates + (int)round.bufs.size() < candidates; ++i) {
      void* p = nullptr;
      // (every fourth candidate one plain allocation: views mapped from 2 MB chunks are the
      // evenly served kind on most boxes — profiles/r05_alloc_method.md — but on one box of
      // round 6 all three were the slow kind and one of two plain allocations the fast one,
      // 78 against 97 us in the bare loop: profiles/r06_fill_geometry.md)
      const int index = rep.candidates + (int)round.bufs.size();
      if (mp_alloc_output(e->device, bytes, index % 4 == 3 ? 0 : (2u << 20), &p) != MP_OK) {
        // out of memory (or of address space): the probe goes on with what there is, and says so
        ++rep.out_of_memory;
        exhausted = true;
        break;
      }
      round.bufs.push_back(p);
    }
    if (round.bufs.empty()) break;
    const int round_first = rep.candidates;
    for (void* p : round.bufs) {
      e->bound[kind] = p;
      double us = 0;
      bool stepped = false;
      rc = tune_impl(e, &us, &stepped, /*quick=*/true);
      if (rc != MP_OK) break;
      if (first) { rep.stepped = stepped ? 1 : 0; first = false; }
      rep.us[rep.candidates] = (float)us;
      if (us < best_us) {
        void* old = round.keep;
        best_us = us; round.keep = p; rep.picked = rep.candidates;
        if (old && std::find(round.bufs.begin(), round.bufs.end(), old) == round.bufs.end())
          (void)free_output(e->device, old, true);
      }
      ++rep.candidates;
    }
    if (rc != MP_OK) break;
    round.release_losers();
    if (rep.candidates - round_first >= 4) {
      // a round whose candidates all take the same time: there is no lottery to win for this
      // view on this box (WORLD.RGB mostly) — stop
      const float lo = *std::min_element(rep.us + round_first, rep.us + rep.candidates);
      const float hi = *std::max_element(rep.us + round_first, rep.us + rep.candidates);
      if (hi < 1.03f * lo) { rep.early_exit = 1; break; }
    }
    // an outlier among them (a fast placement is 8 % or more below the median)?  enough
    if (rep.candidates >= 4) {
      std::vector<float> v(rep.us, rep.us + rep.candidates);
      std::sort(v.begin(), v.end());
      if (v[0] < 0.92f * v[v.size() / 2]) { rep.early_exit = 2; break; }
    }
  }
  if (rc == MP_OK && !round.keep)
    rc = fail(MP_ERR_HIP, "mp_place_output: no buffer of %llu bytes could be mapped: %s",
              (unsigned long long)bytes, g_error.c_str());
  if (rc != MP_OK) return rc;   // (~Round releases everything and rebinds what was bound)
  e->bound[kind] = round.keep;
  rc = mp_tune(e, nullptr);   // the plan for the buffer that stays
  if (rc != MP_OK) return rc;
  round.done = true;
  *device_ptr = round.keep;
  rep.setup_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (report) *report = rep;
  return MP_OK;
}

int mp_box_fill(MpEngine* e, MpObsKind kind, int32_t reps, MpBoxFill* out) {
  if (!e || !out) return fail(MP_ERR_INVALID, "mp_box_fill: NULL argument");
  memset(out, 0, sizeof *out);
  if (kind != MP_OBS_RGB && kind != MP_OBS_WORLD_RGB)
    return fail(MP_ERR_INVALID, "mp_box_fill: kind %d is not a pixel view", (int)kind);
  if (e->ring[kind].base || !e->bound[kind])
    return fail(MP_ERR_INVALID, "mp_box_fill: kind %d is not bound to one buffer", (int)kind);
  if (reps < 1) reps = 1;
  if (reps > 1000) reps = 1000;
  HIP_TRY(hipSetDevice(e->device));
  uint8_t* view = (uint8_t*)e->bound[kind];
  const uint64_t bytes = mp_obs_bytes(e, kind);
  if (bytes == 0 || bytes % 16)
    return fail(MP_ERR_UNSUPPORTED, "mp_box_fill: a view of %llu bytes", (unsigned long long)bytes);
  // the store loop's geometry = the frame launch's under the plan that is current for what is bound
  const bool both = e->bound[MP_OBS_RGB] && e->bound[MP_OBS_WORLD_RGB];
  const int views = both ? 2 : kind == MP_OBS_WORLD_RGB ? 1 : 0;
  const FramePlan& p = e->plan[1][views];
  int waves = p.nwaves - p.feeders;
  if (both) waves = kind == MP_OBS_WORLD_RGB ? p.world_waves : waves - p.world_waves;
  if (waves < 1) waves = 1;
  const int row_cells = kind == MP_OBS_WORLD_RGB ? e->t.W : e->t.vl + e->t.vr + 1;
  const int R = 64 / row_cells > 0 ? 64 / row_cells : 1;
  const uint32_t span = (uint32_t)R * 8u * (uint32_t)row_cells * 24u;   // one renderer pass
  const uint64_t per_world = bytes / (uint64_t)e->N;
  const uint64_t own = (uint64_t)p.ks * (uint64_t)p.B * per_world;       // a workgroup's worlds
  const int groups = p.groups > 0 ? p.groups : 1;
  struct Events {
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  HIP_TRY(hipEventCreate(&ev.a));
  HIP_TRY(hipEventCreate(&ev.b));
  float us[3] = {0, 0, 0};
  for (int what = 0; what < 3; ++what) {
    auto launch = [&]() -> hipError_t {
      if (what == 0) return hipMemsetAsync(view, 0x5a, (size_t)bytes, e->stream);
      const uint64_t owned = what == 1 ? ((own + 15) & ~(uint64_t)15) : 0;
      // (a pooled plan owns less than the view: its rest is walked by the same workgroups)
      const uint64_t cover = what == 1 ? (bytes + (uint64_t)groups - 1) / (uint64_t)groups : 0;
      const uint64_t share = what == 1 ? (owned * (uint64_t)groups >= bytes ? owned : ((cover + 15) & ~(uint64_t)15)) : 0;
      hipLaunchKernelGGL(k_box_fill, dim3(groups), dim3(waves * 64), 0, e->stream, view, bytes, share,
                         what == 1 ? span : 4096u, what == 1 ? 0 : 1);
      return hipGetLastError();
    };
    HIP_TRY(launch());
    HIP_TRY(launch());
    HIP_TRY(hipEventRecord(ev.a, e->stream));
    for (int r = 0; r < reps; ++r) HIP_TRY(launch());
    HIP_TRY(hipEventRecord(ev.b, e->stream));
    HIP_TRY(hipEventSynchronize(ev.b));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
    us[what] = ms * 1e3f / (float)reps;
  }
  out->bytes = bytes;
  out->memset_us = us[0];
  out->product_order_us = us[1];
  out->front_4k_us = us[2];
  out->groups = groups;
  out->waves = waves;
  out->span_bytes = span;
  return sync_and_check(e, "mp_box_fill");
}

int mp_fault_words(const MpEngine* e, uint32_t out[64]) {
  if (!e || !out) return fail(MP_ERR_INVALID, "mp_fault_words: NULL argument");
  for (int i = 0; i < 64; ++i) out[i] = ((const volatile uint32_t*)e->h_fault)[i];
  return MP_OK;
}

int mp_sync(MpEngine* e) {
  if (!e) return fail(MP_ERR_INVALID, "mp_sync: NULL engine");
  HIP_TRY(hipSetDevice(e->device));
  if (int rc = sync_and_check(e, "mp_sync")) return rc;
  return MP_OK;
}

}  // extern "C"
