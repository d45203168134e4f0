// step_common.h — device building blocks shared by the per-substrate step
// functions (one wavefront per world, world record resident in LDS).
//
// These restate the substrate-independent half of the reference's per-step
// path: the avatar components (lua/modules/avatar_library.lua: Avatar move
// updater :155-203, Zapper :570-763) and the grid-engine events they queue
// (moves, beams, teleportToGroup; docs/advanced.md:33-52), in the wavefront
// form described at the top of step_clean_up.h.
//
// Everything here is WAVE-level code: one wave steps one world, and nothing
// synchronises with any other wave (no s_barrier), so the same functions run
// in the one-wave workgroups of the stand-alone step kernels (step_*.hip) and
// inside the 16-wave workgroups of the fused step + render kernel (frame.hip),
// where a few waves step worlds while the others render.
//
// What keeps the dependent chain of a step short (it is latency-, not
// throughput-bound: profiles/r01_end_sq_counters.md):
//   * a lane exchange whose source lane is wave-uniform is a v_readlane (a few
//     cycles), not a ds_bpermute (an LDS round trip): the ordered loops over
//     avatars (moves, respawns, shuffles, sums) run on the scalar unit;
//   * the Fisher-Yates shuffles of a frame are applied to a 16-nibble
//     permutation held in a scalar register pair;
//   * a beam cell fetches the plane bytes of four layers first and their table
//     entries second — two LDS round trips per batch instead of two per layer;
//   * every table that is indexed per lane sits in LDS (`Tables`) or in
//     registers (the site lists of the substrates), loaded next to the record.
#ifndef MP_STEP_COMMON_H_
#define MP_STEP_COMMON_H_

#include "mp_common.h"

namespace stepk {


// Orders the LDS accesses of this wave's lanes: what lanes wrote before is
// visible to every lane after.  DS operations of one wave execute in issue
// order, so this only has to drain the counter and stop the compiler from
// moving or caching LDS accesses across it.
__device__ inline void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Value of `v` in lane `l`; `l` must be wave-uniform.
__device__ inline int rdlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ inline double rdlane(double v, int l) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Unit step of a compass direction, N E S W = 0 1 2 3, N = decreasing y
// (component_library.lua:379-386): {0, 1, 0, -1} / {-1, 0, 1, 0}.  Arithmetic,
// not a table: a per-lane index into a constant array is a memory load.
__device__ inline int dir_dx(int o) { return (o == 1) - (o == 3); }
__device__ inline int dir_dy(int o) { return (o == 2) - (o == 0); }

// This lane's cell of a beam footprint (lane = beam * n + cell), read once per
// step.  BeamShape lives in the kernel arguments: its 16 packed cells are read
// as scalars (one s_load) and picked per lane with masks.  (A loop over cells
// costs a scalar-load round trip per iteration — 2 K cycles a step when measured;
// a per-lane index, or a select chain the compiler turns into one, makes it keep
// a private-memory copy of the whole argument struct.)
struct BeamLane { int nc, lat, fw; uint32_t pred; int per, bl, j; };   // bl = lane / nc, j = lane % nc
__device__ inline BeamLane beam_lane(const BeamShape& shape, int lane) {
  BeamLane r;
  r.nc = shape.n; r.per = shape.per;
  r.bl = (int)(((uint32_t)lane * shape.magic) >> 16);
  const int j = lane - r.bl * r.nc;
  r.j = j;
  uint32_t v = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) v |= shape.cell[q] & (0u - (uint32_t)(j == q));
  r.lat = (int)(int8_t)(v & 255u); r.fw = (int)(int8_t)((v >> 8) & 255u); r.pred = v >> 16;
  return r;
}

// ---- LDS images ------------------------------------------------------------
// Read-only tables of a workgroup (bytes [0, tables_bytes) of DevTables::
// step_blob, laid out by mp_create exactly as here):
//   u32 sinfo[256]       state -> BeamBlocker bits (bit h: blocks hit h, h < 24)
//                        | (player whose avatar state it is + 1) << 24
//   u16 spawn[n_spawn]   the respawn group's cells, creation order (padded to 16 B)
//   i8  action[nact][4]  the ACTION_SET rows: move, turn, fire0, fire1
constexpr int kSinfoBytes = 256 * 4;
__host__ __device__ inline int spawn_bytes(int n_spawn) { return (n_spawn * 2 + 15) & ~15; }
__host__ __device__ inline int tables_bytes(const DevTables& t) {
  return kSinfoBytes + spawn_bytes(t.n_spawn) + ((t.nact * 4 + 15) & ~15);
}

// Per-world scratch of one step.
struct Scratch {
  int8_t victim[MP_MAX_PLAYERS][16];  // avatar hit by cell j of avatar b's zap beam
  uint32_t zapped_mask;
  uint32_t ev_count;                  // events:add calls of this launch
  uint32_t ev[MP_EVENT_ROWS - 1];     // type << 16 | a << 8 | b
  uint32_t flags[2];                  // substrate use, zero at begin_step (the_matrix: bit
                                      // 4 p + k = player p destroyed a class k + 1 resource)
  uint32_t pad[1];
  // followed by uint8_t mark[H*W] (substrate use; all zero between steps)
};
static_assert(sizeof(Scratch) % 16 == 0, "Scratch keeps mark[] 16-byte aligned");

__host__ __device__ inline int mark_bytes(const DevTables& t) { return (t.H * t.W + 15) & ~15; }
// scratch + mark of one world being stepped (substrate extras follow)
__host__ __device__ inline int scratch_bytes(const DevTables& t) {
  return (int)sizeof(Scratch) + mark_bytes(t);
}
// What a wave needs to step one world.
struct World {
  uint8_t* rec;            // LDS: grid planes + WorldTail
  const uint32_t* sinfo;   // LDS tables
  const uint16_t* spawn;
  const int8_t* action_rows;
  Scratch* sc;             // LDS per-world scratch
  uint8_t* mark;
  uint8_t* extra;          // LDS substrate scratch (after mark)
  uint8_t* gw;             // the world's record in HBM
  int w, lane;             // global world index, lane of the wave
  // k_frame: the LDS flag that hands the record to the renderers, and the value that says
  // "this batch's" — stored by finish() as soon as the record and the avatars' head are
  // final, AHEAD of the outputs, the events and the write-back (2.5 K cycles of a first
  // world, with every renderer waiting); NULL in the stand-alone kernels
  uint32_t* publish;
  uint32_t publish_value;
  int next_orders;         // StepArgs::next_orders: finish() leaves the next step's orders in the record
};

__device__ inline World make_world(const DevTables& t, uint8_t* rec, const uint8_t* tables,
                                   uint8_t* scratch, uint8_t* state, int w, int lane) {
  World wd;
  wd.rec = rec;
  wd.sinfo = reinterpret_cast<const uint32_t*>(tables);
  wd.spawn = reinterpret_cast<const uint16_t*>(tables + kSinfoBytes);
  wd.action_rows = reinterpret_cast<const int8_t*>(tables + kSinfoBytes + spawn_bytes(t.n_spawn));
  wd.sc = reinterpret_cast<Scratch*>(scratch);
  wd.mark = scratch + sizeof(Scratch);
  wd.extra = wd.mark + mark_bytes(t);
  wd.gw = state + (size_t)w * t.world_stride;
  wd.w = w; wd.lane = lane;
  wd.publish = nullptr; wd.publish_value = 0; wd.next_orders = 0;
  return wd;
}

// events:add(name, 'dict', ...) (MpEventType in include/mp_engine.h).
__device__ inline void push_event(Scratch* sc, int type, int a, int b) {
  const uint32_t i = atomicAdd(&sc->ev_count, 1u);
  if (i < MP_EVENT_ROWS - 1) sc->ev[i] = ((uint32_t)type << 16) | ((uint32_t)a << 8) | (uint32_t)b;
}

// Avatar p's state, held in lane p's registers for the whole step.
struct Av {
  int x = 0, y = 0, ori = 0, alive = 0, ztimer = 0, ctimer = 0, achange = 0;
  double reward = 0.0;
};

__device__ inline bool step_cell(const DevTables& t, int& x, int& y, int dx, int dy) {
  x += dx; y += dy;
  if (t.topology == 1) {  // TORUS
    x = ((x % t.W) + t.W) % t.W; y = ((y % t.H) + t.H) % t.H;
    return true;
  }
  return x >= 0 && x < t.W && y >= 0 && y < t.H;
}

// k-th set bit (ascending site order) of up to four ballot masks.
__device__ inline int kth_site(unsigned long long m0, unsigned long long m1,
                               unsigned long long m2, unsigned long long m3, int k) {
  int base = 0;
  unsigned long long m = m0;
  int pc = __popcll(m0);
  if (k >= pc) { k -= pc; m = m1; base = 64; pc = __popcll(m1);
    if (k >= pc) { k -= pc; m = m2; base = 128; pc = __popcll(m2);
      if (k >= pc) { k -= pc; m = m3; base = 192; } } }
  for (int i = 0; i < k; ++i) m &= m - 1;
  return base + __ffsll((long long)m) - 1;
}

// This launch's action of avatar `lane` (api:discreteActions, api_factory.lua:81):
// the discrete id is one global load, issued next to the record's; the ACTION_SET
// row (discrete_action_wrapper.py:97-109) is looked up in the LDS tables once
// they are there — a second, dependent trip to memory would cost a round trip.
struct Action { int move = 0, turn = 0, fire0 = 0, fire1 = 0, bad = 0; };
// In the raw-field form (mp_step_fields: dmlab2d's own "<player>.<name>" action
// surface, avatar_library.lua:205-223) the lane loads its avatar's nfields
// values and returns them in the ACTION_SET row format, bit 30 set; a field
// outside its actionSpec range makes the whole action a counted NOOP (-1).
constexpr int kFieldsTag = 0x40000000;
__device__ inline int fetch_action_id(const DevTables& t, const int32_t* actions, int mode,
                                      int w, int lane) {
  if (mode == STEP_MODE_RESET || lane >= t.P) return 0;
  if (mode == STEP_MODE_FIELDS) {
    const int32_t* f = actions + ((size_t)w * t.P + lane) * t.nfields;
    int v[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) v[a] = a < t.nfields ? f[a] : 0;
    uint32_t row = 0;
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int lo = (int)(int8_t)(t.field_lo >> (8 * a)), hi = (int)(int8_t)(t.field_hi >> (8 * a));
      ok = ok && (a >= t.nfields || (v[a] >= lo && v[a] <= hi));
      row |= ((uint32_t)v[a] & 255u) << (8 * a);
    }
    return ok ? (int)(row & 0x3fffffffu) | kFieldsTag : -1;
  }
  return actions[(size_t)w * t.P + lane];
}
__device__ inline Action lookup_action(const DevTables& t, const World& wd, int act, int mode) {
  Action r;
  if (mode == STEP_MODE_RESET || wd.lane >= t.P) return r;
  uint32_t row;
  if (mode == STEP_MODE_FIELDS) {
    if (act < 0) { r.bad = 1; return r; }
    row = (uint32_t)act & 0x3fffffffu;   // (fire1 is 0 / 1: bits 30-31 carry nothing)
  } else {
    if (act < 0 || act >= t.nact) { act = 0; r.bad = 1; }
    row = *reinterpret_cast<const uint32_t*>(wd.action_rows + 4 * act);
  }
  r.move = (int)(int8_t)(row & 255u); r.turn = (int)(int8_t)((row >> 8) & 255u);
  r.fire0 = (int)(int8_t)((row >> 16) & 255u); r.fire1 = (int)(int8_t)(row >> 24);
  return r;
}

// A1: the engine visits the pieces of an updater group in a freshly shuffled
// order every frame; forward Fisher-Yates, one draw per position.  Up to four
// orders at once: lane 16 g + i draws position i's partner for stream s<g>; the
// swaps are applied to the 16-nibble identity permutation.  Lane p < P gets, in
// out[g], the avatar that stream g visits p-th.
__device__ inline void shuffled_orders(int lane, int P, int s0, int s1, int s2, int s3, int n,
                                       uint32_t step, uint32_t ep, uint32_t k0, uint32_t k1,
                                       int (&out)[4]) {
  const int g = lane >> 4, pos = lane & 15;
  const int stream = g == 0 ? s0 : g == 1 ? s1 : g == 2 ? s2 : s3;
  int j = pos;
  if (g < n && pos + 1 < P)
    j = pos + (int)philox_bounded(
        philox4x32_10((uint32_t)pos, (uint32_t)stream, step, ep, k0, k1), (uint32_t)(P - pos));
  // Every lane of group g carries group g's 16-nibble permutation, so the swap chains of all
  // the streams run in the SAME instructions (round 4, second session: held in scalar register
  // pairs and swapped stream by stream — 9 dependent scalar instructions a swap, a wave issues
  // one instruction every four cycles — the orders were 2.7 - 3.2 K of a step's 17 - 21 K
  // cycles).  The partner of position i of a lane's own group comes by ds_bpermute, requested
  // one iteration ahead.  Same draws, same swaps in the same order per stream.
  unsigned long long perm = 0xFEDCBA9876543210ull;
  const int row = lane & 48;
  int ji = __shfl(j, row);
  for (int i = 0; i + 1 < P; ++i) {
    const int jn = __shfl(j, row | ((i + 1) & 15));
    const unsigned long long x = ((perm >> (4 * i)) ^ (perm >> (4 * ji))) & 15ull;
    perm ^= (x << (4 * i)) | (x << (4 * ji));
    ji = jn;
  }
  const int mine = (int)((perm >> (4 * pos)) & 15ull);   // stream g visits avatar `mine` pos-th
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = __shfl(mine, 16 * q + pos);
}

// The streams a substrate shuffles per frame (what shuffled_orders takes), so that finish()
// can work out the NEXT step's orders.
struct OrderStreams { int s0, s1, s2, s3, n; };

// The orders of step `step`: read from the record when the previous step's finish() left
// them there (WorldTail::orders_step), drawn here otherwise — the same values either way:
// they depend on (stream, position, step, episode, seed, P) and on nothing the step does.
// 2.1 K of a clean_up step's 13 K cycles in front of the hand-over (profiles/r04_head.md).
__device__ inline void step_orders(const WorldTail* tail, int lane, int P, const OrderStreams& os,
                                   uint32_t step, uint32_t ep, uint32_t k0, uint32_t k1,
                                   int (&out)[4]) {
  // (step 0 is the grid:update of api:start — territory runs its updaters there too — and 0 is
  // the tag's "none": the orders of a reset are always drawn)
  if (step != 0u && __builtin_amdgcn_readfirstlane((int)tail->orders_step) == (int)step) {
    const uint32_t v = tail->next_orders[lane & 15];
#pragma unroll
    for (int q = 0; q < 4; ++q) out[q] = (int)((v >> (4 * q)) & 15u);
    return;
  }
  shuffled_orders(lane, P, os.s0, os.s1, os.s2, os.s3, os.n, step, ep, k0, k1, out);
}

// ---- record / table movement -------------------------------------------------
// Pins a loaded value where it is: without it the compiler sinks each load of a
// batch into the conditional store that consumes it, and a copy meant to have
// eight loads in flight pays eight memory round trips one after the other (what
// round 1's kernels did: 15 K of a step's 37 K cycles).
__device__ inline void issued(uint4& v) {
  asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

// World record HBM -> LDS.  Every load is issued before the first one is waited
// for: a load-store loop would pay one HBM round trip per 1 KiB.  No sync.
__device__ inline void load_record(const DevTables& t, uint8_t* rec, const uint8_t* gw, int lane) {
  const int nvec = t.world_stride >> 4;
  const uint4* src = reinterpret_cast<const uint4*>(gw);
  uint4* dst = reinterpret_cast<uint4*>(rec);
  for (int i0 = 0; i0 < nvec; i0 += 8 * 64) {
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * 64 + lane;
      v[k] = src[i < nvec ? i : nvec - 1];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) issued(v[k]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * 64 + lane;
      if (i < nvec) dst[i] = v[k];
    }
  }
}

// The workgroup's read-only tables (DevTables::step_blob) -> LDS, by `nthreads`
// threads.  No sync.
__device__ inline void load_tables(const DevTables& t, uint8_t* tables, int tid, int nthreads) {
  const int nvec = tables_bytes(t) >> 4;
  for (int i = tid; i < nvec; i += 2 * nthreads) {   // (<= 1.5 KB: one or two rounds)
    uint4 a = reinterpret_cast<const uint4*>(t.step_blob)[i];
    uint4 b = reinterpret_cast<const uint4*>(t.step_blob)[min(i + nthreads, nvec - 1)];
    issued(a); issued(b);
    reinterpret_cast<uint4*>(tables)[i] = a;
    if (i + nthreads < nvec) reinterpret_cast<uint4*>(tables)[i + nthreads] = b;
  }
}

// Zeroes the marks (once per LDS allocation: every step leaves them zero).
__device__ inline void clear_marks(const DevTables& t, uint8_t* mark, int lane) {
  const int nvec = mark_bytes(t) >> 4;
  for (int i = lane; i < nvec; i += 64) reinterpret_cast<uint4*>(mark)[i] = uint4{0, 0, 0, 0};
}

__device__ inline void begin_step(Scratch* sc, int lane) {
  if (lane == 0) { sc->ev_count = 0; sc->zapped_mask = 0; sc->flags[0] = 0; sc->flags[1] = 0; }
}

// Clears n bytes at byte offset off of the record (a whole plane: beam sprites
// of the previous frame).  Planes start at layer * H * W: 2-byte stores when that
// is even (every map so far), bytes otherwise; no branches per element.
__device__ inline void clear_bytes(uint8_t* rec, int off, int n, int lane) {
  if (((off | n) & 1) == 0) {
    for (int i = 2 * lane; i < n; i += 128) *reinterpret_cast<uint16_t*>(rec + off + i) = 0;
  } else {
    for (int i = lane; i < n; i += 64) rec[off + i] = 0;
  }
}

__device__ inline void load_avatars(const WorldTail* tail, int lane, Av& a) {
  if (lane < MP_MAX_PLAYERS) {
    a.x = tail->ax[lane]; a.y = tail->ay[lane]; a.ori = tail->aori[lane];
    a.alive = tail->aalive[lane]; a.ztimer = tail->ztimer[lane];
    a.ctimer = tail->ctimer[lane]; a.achange = tail->achange[lane];
  }
}

// World build of an episode with 'choice' map characters: the init grid holds
// the optional objects; all of them are taken off, then those that exist in the
// outcome of their choice are put (back): alternatives of one character may share
// a cell-layer (the_matrix's resource classes).  Outcome of choice c = draw
// (RS_MAP_CHOICE, index c) bounded by its list length (prefab_utils.lua:101-103:
// random:choice(prefab.list)).  Row = cell, plane | initial state << 8, choice,
// outcome mask.
__device__ inline void apply_map_choices(const DevTables& t, uint8_t* grid, int lane,
                                         uint32_t ep, uint32_t k0, uint32_t k1) {
  const int HW = t.H * t.W;
  for (int i = lane; i < t.n_optional; i += 64) {
    const int4 o = reinterpret_cast<const int4*>(t.optional)[i];
    grid[(o.y & 255) * HW + o.x] = 0;
  }
  wsync();
  for (int i = lane; i < t.n_optional; i += 64) {
    const int4 o = reinterpret_cast<const int4*>(t.optional)[i];
    // choice word: id | first outcome this row's mask covers << 16; a negative
    // outcome count = drawn once per WORLD (coins.py draws its map in build()):
    // the draw then does not carry the episode
    const int cid = o.z & 0xffff, base = o.z >> 16;
    int n = t.choice_n[cid];
    uint32_t epw = ep;
    if (n < 0) { n = -n; epw = 0xffffffffu; }
    const int k = (int)philox_bounded(
        philox4x32_10((uint32_t)cid, RS_MAP_CHOICE, 0u, epw, k0, k1), (uint32_t)n) - base;
    if (k >= 0 && k < 32 && ((o.w >> k) & 1)) grid[(o.y & 255) * HW + o.x] = (uint8_t)(o.y >> 8);
  }
  wsync();
}

// Episode start of the avatars: _avatarStart (base_simulation.lua:396-445) —
// per initial spawn group a partial Fisher-Yates over the group's cells in
// creation order, avatar i taking the next sampled cell of its group — and
// Avatar:start (avatar_library.lua:288-320), random:choice(_COMPASS).
// (Episode start only: the table reads below stay in global memory.)
__device__ inline void spawn_avatars(const DevTables& t, uint8_t* grid, int lane,
                                     uint32_t ep, uint32_t k0, uint32_t k1, Av& a,
                                     int alive_state = -1) {   // (-1: the pack's)
  const int P = t.P, HW = t.H * t.W;
  const bool is_av = lane < P;
  const int my_group = is_av ? t.avatar_init_group[lane] : -1;
  int my_cell = 0;
  for (int g = 0; g < t.n_init_groups; ++g) {
    const int base = t.init_spawn_ptr[g];
    int ns = t.init_spawn_ptr[g + 1] - base;   // <= 128: two cells per lane
    const unsigned long long members = __ballot(my_group == g);
    const int want = __popcll(members);
    int item = lane < ns ? t.init_spawn_cells[base + lane] : 0;
    int item_hi = 64 + lane < ns ? t.init_spawn_cells[base + 64 + lane] : 0;
    if (t.optional_spawn) {
      // a spawn point of a 'choice' character may not exist this episode: it does
      // iff a piece of the spawn group stands on its cell; the pool is the present
      // cells in creation order (mp_create: at most 64 cells in such a group)
      bool present = false;
      if (lane < ns)
        for (int l = 0; l < t.L; ++l) {
          const int s = grid[l * HW + item];
          if (s != 0 && (t.state_groups[s] & t.init_spawn_mask[g]) != 0) present = true;
        }
      const unsigned long long pm = __ballot(present);
      const int dst = __popcll(pm & ((1ull << lane) - 1ull));
      int packed = 0;
      for (int i = 0; i < ns; ++i) {   // compaction: lane dst takes lane i's cell
        const int ci = __shfl(item, i), di = __shfl(dst, i);
        if (((pm >> i) & 1ull) && lane == di) packed = ci;
      }
      item = packed;
      ns = __popcll(pm);
    }
    int j = lane;
    if (lane < want && lane < ns)   // (fewer points than avatars: mp_create refuses such packs)
      j = lane + (int)philox_bounded(
          philox4x32_10((uint32_t)(lane + 256 * g), RS_START_SPAWN, 0u, ep, k0, k1),
          (uint32_t)(ns - lane));
    for (int i = 0; i < want; ++i) {   // (want <= 16: position i lives in `item` of lane i)
      const int ji = __shfl(j, i);
      const int vi = __shfl(item, i);
      const int vj = ji < 64 ? __shfl(item, ji) : __shfl(item_hi, ji - 64);
      if (ji == i) continue;
      if (lane == i) item = vj;
      if (ji < 64) { if (lane == ji) item = vi; }
      else if (lane == ji - 64) item_hi = vi;
    }
    const int rank = __popcll(members & ((1ull << lane) - 1ull));
    const int got = __shfl(item, my_group == g ? rank : 0);
    if (my_group == g) my_cell = got;
  }
  a = Av();
  if (is_av) {
    a.ori = (int)philox_bounded(
        philox4x32_10((uint32_t)lane, RS_START_ORIENT, 0u, ep, k0, k1), 4u);
    a.x = my_cell % t.W; a.y = my_cell / t.W; a.alive = 1;
    grid[t.avatar_layer * HW + my_cell] =
        (uint8_t)(alive_state >= 0 ? alive_state : t.alive_state[lane]);
  }
}

// Avatar move updater (avatar_library.lua:155-203) + the queued turn / moveRel
// events, resolved in the frame's visiting order with one ballot per avatar.
// Returns `wants` (this lane's avatar attempted a move) — the caller fires the
// onContact enter on the avatar's final cell for those (A3b: a blocked move
// re-enters in place).  `alive_state` is this lane's avatar state id.  The grid
// reflects the moves afterwards.
__device__ inline bool resolve_moves(const DevTables& t, const World& wd, Av& a, int a_move,
                                     int a_turn, int order_move, int alive_state,
                                     int follow_layer = -1, bool has_follower = true) {
  uint8_t* grid = wd.rec;
  const int lane = wd.lane;
  const int P = t.P, HW = t.H * t.W, W = t.W;
  const bool is_av = lane < P;
  if (is_av && a_turn != 0) a.ori = (a.ori + a_turn + 4) & 3;  // off-grid pieces turn too
  const bool wants = is_av && a.alive && a_move != 0;
  int tx = a.x, ty = a.y;
  bool target_free = false;  // in bounds and no static piece on the avatar layer
  wsync();                   // earlier grid writes are visible
  if (wants) {
    const int dir = (a.ori + a_move - 1) & 3;
    if (step_cell(t, tx, ty, dir_dx(dir), dir_dy(dir))) {
      const int s = grid[t.avatar_layer * HW + ty * W + tx];
      target_free = s == 0 || (wd.sinfo[s] >> 24) != 0;  // other avatars: decided in order below
      // a connected piece needs its own target free too: an orphaned follower
      // (its avatar is gone) blocks; a live avatar's follower moves with it
      // (`has_follower`: externality_mushrooms' avatars can lose theirs, step_mushroom.h)
      if (follow_layer >= 0 && has_follower && s == 0 && grid[follow_layer * HW + ty * W + tx] != 0)
        target_free = false;
    }
  }
  const int old_cell = a.y * W + a.x;
  bool moved = false;
  const int try_move = (int)(wants && target_free);
  if (__ballot(try_move != 0) != 0) {
    for (int r = 0; r < P; ++r) {
      const int p = rdlane(order_move, r);
      if (rdlane(try_move, p) == 0) continue;
      const int ptx = rdlane(tx, p), pty = rdlane(ty, p);
      const bool occupied = __ballot(is_av && a.alive && a.x == ptx && a.y == pty) != 0;
      if (lane == p && !occupied) { a.x = ptx; a.y = pty; moved = true; }
    }
  }
  // a connected piece (grid:connect, A14) on `follow_layer` moves with the avatar
  int follower = 0;
  const bool follows = moved && follow_layer >= 0 && has_follower;
  if (follows) follower = grid[follow_layer * HW + old_cell];
  if (moved) {
    grid[t.avatar_layer * HW + old_cell] = 0;
    if (follows) grid[follow_layer * HW + old_cell] = 0;
  }
  wsync();
  if (moved) {
    grid[t.avatar_layer * HW + a.y * W + a.x] = (uint8_t)alive_state;
    if (follows) grid[follow_layer * HW + a.y * W + a.x] = (uint8_t)follower;
  }
  wsync();
  return wants;
}

// hitBeam (game_object.lua:246-258) for every avatar with `fire` set, footprint
// of Zapper:getWhoZappable (avatar_library.lua:780-824): lane (b, j) evaluates
// cell j of avatar b's beam; a ballot of the "stops the beam" predicate against
// the cell's predecessor mask gives reached / not reached for all cells at once.
//   extra(state, cell)  -> substrate onHit of the piece in that state: bit 0 =
//                          it returns true (stops the beam), bit 1 = it reacts,
//                          bits 8.. = 1 + index of an avatar the hit is reported
//                          for in sc->victim (a piece standing in for it);
//   on_cells(b0, per, nc, reached, cell, touched)
//                       -> substrate effects, called once per round of beams
//                          (`touched` = reached and some piece reacted).
//   `hit` is the hit's index in the pack (BeamBlocker bit), `only` restricts
//   the call to one avatar's beam (-1 = all): substrates whose beams change
//   state inside the flush evaluate them one at a time, in visiting order.
//   `zmat` (this world's [P][P] block of the zap matrix, or NULL) and `nzapped`
//   (this lane's count of avatars its beam hit, or NULL) are debug observations.
// A4: the beam sprite is drawn on the hit's layer, blocked cell included.
template <class ExtraBlock, class OnCells>
__device__ inline void fire_beams(const DevTables& t, const World& wd, WorldTail* tail,
                                  const Av& a, bool fire, const BeamLane& shape, int hit,
                                  bool zap, int beam_layer, int s_beam, bool remove_hit,
                                  ExtraBlock extra_block, OnCells on_cells, int only = -1,
                                  double* zmat = nullptr, int* nzapped = nullptr) {
  uint8_t* grid = wd.rec;
  Scratch* sc = wd.sc;
  const int lane = wd.lane;
  const int P = t.P, HW = t.H * t.W, W = t.W, L = t.L;
  const int nc = shape.nc;
  const int per = shape.per;  // beams per round (64 / nc)
  const int firing = (int)(fire && a.alive);
  // Nobody fires (clean_up under uniformly random play: 44 % of the steps, for either beam):
  // nothing below has an effect — no sprite, no victim anyone reads (zap_rewards looks at
  // firing owners only), no callback that acts — and it was 1.7 - 2.0 K cycles of shuffles,
  // ballots and LDS round trips per call.
  if (__ballot(firing != 0 && (only < 0 || lane == only)) == 0ull) {
    wsync();
    return;
  }
  for (int b0 = 0; b0 < P; b0 += per) {
    const int bl = shape.bl, j = shape.j, b = b0 + bl;
    const bool lane_ok = bl < per && b < P;
    const int bs = lane_ok ? b : 0;
    const bool bfire = __shfl(firing, bs) != 0 && lane_ok && (only < 0 || b == only);
    const int bx = __shfl(a.x, bs), by = __shfl(a.y, bs), bo = __shfl(a.ori, bs);
    // cell = pos + lat * right(bo) + fwd * forward(bo)
    const int lat = shape.lat, fw = shape.fw;
    const int rdir = (bo + 1) & 3;
    int x = bx, y = by;
    const bool inb = step_cell(t, x, y, lat * dir_dx(rdir) + fw * dir_dx(bo),
                               lat * dir_dy(rdir) + fw * dir_dy(bo));
    const int cell = inb ? y * W + x : 0;
    bool blocked = false, extra_hit = false;
    int hit_player = -1;
    if (bfire && inb) {
      // four planes at a time: all plane bytes first, all table entries second —
      // two LDS round trips per batch instead of two per layer
      for (int l0 = 0; l0 < L; l0 += 4) {
        uint32_t st[4], info[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) st[k] = l0 + k < L ? grid[(l0 + k) * HW + cell] : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) info[k] = wd.sinfo[st[k]];   // sinfo[0] == 0
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int s = (int)st[k];
          if (s == 0) continue;
          // BeamBlocker:onHit (component_library.lua:678-685)
          if ((info[k] >> hit) & 1u) blocked = true;
          const int pl = (int)(info[k] >> 24) - 1;
          // Zapper:onHit (avatar_library.lua:652-681); on-grid => alive
          if (pl >= 0 && zap) { hit_player = pl; blocked = true; }
          const int xb = extra_block(s, cell);
          if (xb & 1) blocked = true;
          if (xb & 2) extra_hit = true;
          if (xb >> 8) hit_player = (xb >> 8) - 1;
        }
      }
    }
    // every ray stops at the first cell that is outside the map or blocks
    const unsigned long long stops = __ballot(bfire && (!inb || blocked));
    const uint32_t mine = (uint32_t)(stops >> (bl * nc)) & ((1u << nc) - 1u);
    const bool reached = bfire && inb && (mine & shape.pred) == 0;
    if (reached) grid[beam_layer * HW + cell] = (uint8_t)s_beam;
    const bool zhit = reached && hit_player >= 0;
    if (zhit && remove_hit) atomicOr(&sc->zapped_mask, 1u << hit_player);
    if (zhit && zap) push_event(sc, MP_EVENT_ZAP, b + 1, hit_player + 1);
    // playerZapMatrix(zapped, zapper) (avatar_library.lua:657-659): a beam meets
    // a given avatar at most once
    if (zhit && zap && zmat) zmat[hit_player * P + b] = 1.0;
    if (lane_ok) sc->victim[b][j] = (int8_t)(zhit ? hit_player : -1);
    const unsigned long long zb = __ballot(zhit);
    if (lane == 0 && zb != 0) tail->ctr[4] += __popcll(zb);
    // Zapper.num_others_player_zapped_this_step of the beam's owner (:672-677)
    if (nzapped && lane >= b0 && lane < b0 + per && lane < P)
      *nzapped += __popcll((zb >> ((lane - b0) * nc)) & ((1ull << nc) - 1ull));
    on_cells(b0, per, nc, reached, cell, reached && extra_hit);
  }
  wsync();
}

// Zapper:onHit rewards in the reference's event order (zap visiting order, then
// footprint order) so that the f64 sums are bit-identical.
// INVARIANT (fire_beams' early return relies on it): `fire_zap` is the very flag the
// preceding fire_beams call was given, and it is only ever set for an avatar that was alive
// when it was set (every level sets it under `if (a.alive && ...)`), so fire_beams' own
// `fire && a.alive` is the same predicate.  victim[owner][*] is read for owners with the flag
// set only; when no lane has it set fire_beams returns before it writes victim[][] — the
// scratch then still holds ANOTHER world's victims — and nothing here reads them.  Do not
// re-test a.alive here: a firing avatar zapped in the same frame still pays and collects
// (Zapper:onHit fires for every beam of the frame, avatar_library.lua:652-681).
__device__ inline void zap_rewards(const DevTables& t, const Scratch* sc, int lane, Av& a,
                                   bool fire_zap, int order_zap, int nc, double penalty,
                                   double reward) {
  if (penalty == 0.0 && reward == 0.0) return;
  const int firing = (int)fire_zap;
  for (int r = 0; r < t.P; ++r) {
    const int owner = rdlane(order_zap, r);
    if (rdlane(firing, owner) == 0) continue;
    for (int q = 0; q < nc; ++q) {
      const int victim = sc->victim[owner][q];
      if (victim < 0) continue;
      if (lane == victim) a.reward += penalty;
      if (lane == owner) a.reward += reward;
    }
  }
}

// Zapper respawn updater + teleportToGroup(spawnGroup, aliveState), PICK_RANDOM
// orientation (avatar_library.lua:638-649, component_library.lua:336-354).
// A5: uniform over the group's pieces in creation order; an occupied target
// fails and is retried next frame.  Returns the respawn cell or -1.
__device__ inline int resolve_respawns(const DevTables& t, const World& wd, WorldTail* tail,
                                       Av& a, bool want_respawn, int order_resp,
                                       int alive_state, uint32_t step, int frame, uint32_t ep,
                                       uint32_t k0, uint32_t k1) {
  uint8_t* grid = wd.rec;
  const int lane = wd.lane;
  const int P = t.P, HW = t.H * t.W, W = t.W;
  const bool is_av = lane < P;
  if (__ballot(want_respawn) == 0) return -1;
  int rcell = 0, rori = 0;
  bool rfree = false;
  if (want_respawn) {
    const Philox4 d = philox4x32_10((uint32_t)lane, RS_RESPAWN, step, ep, k0, k1);
    rcell = wd.spawn[philox_bounded(d, (uint32_t)t.n_spawn)];
    rori = (int)(d.x3 & 3u);
    const int s = grid[t.avatar_layer * HW + rcell];
    rfree = s == 0 || (wd.sinfo[s] >> 24) != 0;
  }
  bool respawned = false;
  const int try_it = (int)(want_respawn && rfree);
  for (int r = 0; r < P; ++r) {
    const int p = rdlane(order_resp, r);
    if (rdlane(try_it, p) == 0) continue;
    const int pc = rdlane(rcell, p);
    const bool occupied = __ballot(is_av && a.alive && a.y * W + a.x == pc) != 0;
    if (lane == p && !occupied) {
      a.alive = 1; a.x = pc % W; a.y = pc / W; a.achange = frame; a.ori = rori;
      respawned = true;
    }
  }
  if (respawned) grid[t.avatar_layer * HW + rcell] = (uint8_t)alive_state;
  const unsigned long long rb = __ballot(respawned);
  if (lane == 0 && rb != 0) tail->ctr[6] += __popcll(rb);
  return respawned ? rcell : -1;
}

// flush 2 of a zap: avatar -> playerWait (off-grid).
__device__ inline void apply_zapped(const DevTables& t, const World& wd, Av& a, bool respawned,
                                    int frame) {
  const uint32_t zapped = wd.sc->zapped_mask;
  if (wd.lane < t.P && a.alive && !respawned && ((zapped >> wd.lane) & 1u)) {
    wd.rec[t.avatar_layer * t.H * t.W + a.y * t.W + a.x] = 0;
    a.alive = 0; a.achange = frame;
  }
}

// Avatar registers -> record, the per-player / per-world outputs, and the
// record back to HBM.
__device__ inline void finish(const DevTables& t, const World& wd, WorldTail* tail, const Av& a,
                              double aux0, int zap_cooldown, int step_type,
                              const StepOutputs& out, const OrderStreams& os) {
  const int P = t.P, lane = wd.lane, w = wd.w;
  if (lane < MP_MAX_PLAYERS) {
    tail->ax[lane] = (uint8_t)a.x; tail->ay[lane] = (uint8_t)a.y;
    tail->aori[lane] = (uint8_t)a.ori; tail->aalive[lane] = (uint8_t)a.alive;
    tail->ztimer[lane] = (uint8_t)a.ztimer; tail->ctimer[lane] = (uint8_t)a.ctimer;
    tail->achange[lane] = a.achange;
  }
  if (wd.publish) {   // the renderers read the planes and the head written just above
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0)
      __hip_atomic_store(wd.publish, wd.publish_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (lane < P) {
    // "N.REWARD", "N.READY_TO_SHOOT" (avatar_library.lua:737-744), substrate metric
    const size_t o = (size_t)w * P + lane;
    out.reward[o] = a.reward;
    const double v = 1.0 - (double)a.ztimer / (double)zap_cooldown;
    out.ready[o] = a.alive ? (v > 0.0 ? v : 0.0) : 0.0;
    out.aux0[o] = aux0;
    out.position[o * 2 + 0] = a.x;
    out.position[o * 2 + 1] = a.y;
    out.orientation[o] = a.ori;
  }
  // COLLECTIVE_REWARD = sum over players in index order (collective_reward_wrapper.py:49)
  double sum = 0.0;
  for (int p = 0; p < P; ++p) sum += rdlane(a.reward, p);
  if (lane == 0) {
    out.collective[w] = sum;
    out.step_type[w] = step_type;
    out.discount[w] = step_type == 1 ? 1.0 : 0.0;
    tail->reward_fx += (int32_t)(sum * 1024.0);
  }
  wsync();
  {
    // api:events: header row + one row per event (unused rows are not written)
    const Scratch* sc = wd.sc;
    const uint32_t total = sc->ev_count;
    const uint32_t n = total < MP_EVENT_ROWS - 1 ? total : MP_EVENT_ROWS - 1;
    int4* rows = reinterpret_cast<int4*>(out.events) + (size_t)w * MP_EVENT_ROWS;
    if (lane == 0) rows[0] = int4{(int)n, (int)(total - n), 0, 0};
    for (uint32_t i = (uint32_t)lane; i < n; i += 64) {
      const uint32_t e = sc->ev[i];
      rows[1 + i] = int4{(int)(e >> 16), (int)((e >> 8) & 255u), (int)(e & 255u), 0};
    }
  }
  // the next step's shuffled orders, while nobody waits for this wave (step_orders)
  if (wd.next_orders && os.n > 0 && !__builtin_amdgcn_readfirstlane(tail->done)) {
    const uint32_t next = (uint32_t)__builtin_amdgcn_readfirstlane(tail->step) + 1u;
    const uint32_t ep = (uint32_t)__builtin_amdgcn_readfirstlane((int)tail->episode) - 1u;
    const uint64_t seed = tail->seed;
    int o[4];
    shuffled_orders(lane, P, os.s0, os.s1, os.s2, os.s3, os.n, next, ep,
                    (uint32_t)seed, (uint32_t)(seed >> 32), o);
    if (lane < MP_MAX_PLAYERS)
      tail->next_orders[lane] = (uint16_t)(o[0] | (o[1] << 4) | (o[2] << 8) | (o[3] << 12));
    if (lane == 0) tail->orders_step = next;
  } else if (lane == 0) {
    tail->orders_step = 0u;
  }
  wsync();
  const int nvec = t.world_stride >> 4;
  for (int i0 = 0; i0 < nvec; i0 += 8 * 64) {
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * 64 + lane;
      v[k] = reinterpret_cast<const uint4*>(wd.rec)[i < nvec ? i : nvec - 1];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) issued(v[k]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * 64 + lane;
      if (i < nvec) reinterpret_cast<uint4*>(wd.gw)[i] = v[k];
    }
  }
}

// Decides reset / step / frozen for this launch (wave-uniform).
//   returns 0: nothing to do, 1: reset, 2: step
__device__ inline int dispatch(const DevTables& t, const WorldTail* tail, int lane, int w,
                               const uint8_t* reset_mask, int mode, int auto_reset,
                               const StepOutputs& out) {
  if (mode == STEP_MODE_RESET) return (reset_mask ? reset_mask[w] != 0 : true) ? 1 : 0;
  if (!tail->started) return 0;  // never reset: nothing to step
  if (tail->done && auto_reset) return 1;
  if (tail->done) {              // frozen after LAST until mp_reset
    if (lane < t.P) out.reward[w * t.P + lane] = 0.0;
    if (lane == 0) {
      out.collective[w] = 0.0; out.step_type[w] = 2; out.discount[w] = 0.0;
      reinterpret_cast<int4*>(out.events)[(size_t)w * MP_EVENT_ROWS] = int4{0, 0, 0, 0};
    }
    return 0;
  }
  return 2;
}

// What a step kernel is launched with besides the tables.
struct StepArgs {
  uint8_t* state;
  const int32_t* actions;
  const uint8_t* reset_mask;
  int mode, auto_reset, num_worlds;
  int next_orders;   // 1: finish() leaves the next step's shuffled orders in the record
  StepOutputs out;
};

// Substrates without LDS extras behind the marks (territory overloads both).
template <class Tables>
__host__ __device__ inline int extra_bytes(const Tables&) { return 0; }
template <class Tables>
__device__ inline void init_extra(const DevTables&, const Tables&, uint8_t*, int) {}

}  // namespace stepk

#endif  // MP_STEP_COMMON_H_
