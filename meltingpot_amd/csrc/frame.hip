// frame.hip — the observation kernels: layer views + tile renderer for N worlds,
// persistent, with the environment step fused in.
//
// Replaces the observation reads that follow every reference step
// (api:observation, lua/modules/api_factory.lua:73-75):
//   "N.RGB"      playerLayerView:observation -> playerView:render
//                (avatar_library.lua:225-277)   egocentric 11x11 cells, 88x88x3
//   "WORLD.RGB"  worldView:render(worldLayerView:observation)
//                (base_simulation.lua:347-368)  whole map, H*8 x W*8 x 3
// i.e. dmlab2d's `world:createView` + `tile.Scene:render` — and, in the fused
// form, api:advance itself (api_factory.lua:104-111; step_<substrate>.h).
// Semantics of the views (window, rotation, OutOfBounds, per-viewer spriteMap,
// relative facing, bottom->top 8-bit alpha compositing) are the ones the CPU
// restatement in oracle/render.c documents as assumptions A6-A9; this file is
// bit-exact with it.
//
// Execution shape (v12).  The kernel writes 192 B per output cell and reads
// ~9 B, so it is HBM-write bound by construction; what the store path charges
// for on MI355X is the NUMBER of vector store instructions a wave issues
// (profiles/r01_render_ablation.md), so the observation leaves as few, full,
// 16-byte-per-lane stores as possible.  Round 1 (v11) ran one launch per step
// for the rules (one wave per world, 25-90 us, latency-bound: SQ_WAIT_ANY 60 %)
// and one for the pixels whose workgroups each paid a 14-17 us prologue.  v12 is
// ONE persistent launch per bound view:
//   * grid = one workgroup per CU (all 160 KB of LDS; 12 waves for WORLD.RGB,
//     16 for the per-agent views: plan_frame); a workgroup owns a contiguous
//     range of worlds and walks it in batches of B worlds through two LDS
//     record buffers;
//   * the workgroup's prologue stages what never changes — the blob with the
//     de-duplicated sprite atlas, the composite cache and the lookup tables,
//     plus the step's tables — once per CU instead of once per 8 worlds;
//   * the last F waves are FEEDERS, the others RENDERERS (two code paths of one
//     kernel; the feeders run at raised wave priority: a step is a chain of
//     dependent instructions, the renderers always have independent work).
//     A feeder brings worlds of the next batch into the free buffer — record
//     HBM -> LDS, then (fused form) the whole environment step on it, in LDS,
//     by that one wave (step_<substrate>.h), and the stepped record streamed
//     back to HBM — while the renderers draw the current batch.  The step's
//     dependent-latency chain (~10 us of LDS round trips and scalar waits,
//     almost no issue slots) hides behind the store-bound rendering of the
//     previous batch; only the first batch of a workgroup is exposed;
//   * rendering is handed out as tickets (batch, pass) from one LDS counter; a
//     renderer with a ticket waits (s_sleep polling of LDS flags) until every
//     world of that batch has been fed, so there is no workgroup barrier after
//     the prologue; a feeder refills a buffer once every pass of its previous
//     batch has been counted done;
//   * work unit = a "strip": one row of output cells = 8 pixel rows, contiguous
//     in the output tensor in both views.  A pass = floor(64 / row_cells) whole
//     strips = one contiguous 64-byte-aligned span;
//   * phase 1, one lane per cell: resolve the cell's draw list from the LDS
//     planes — top -> bottom, stopping at the first fully opaque sprite.  Stacks
//     that static pieces of the map form (dirt on water, a shadow on sand,
//     claimed-resource paint on its texture) were pre-blended by mp_create: a
//     hash lookup replaces (base, overlay) by the composite image, so only
//     avatars, beams and rare combinations are left to composite.  Those cells
//     are listed densely (8-bit alpha ones first);
//   * phase 2b, eight lanes per listed cell (one per pixel row): binary-alpha
//     select or the 8-bit blend in registers, result staged in the wave's LDS
//     scratch as one more pre-packed image;
//   * phase 2a: the span leaves as 16-byte lane-contiguous chunks (1 KiB per
//     wave store, whole cache lines); each half chunk is an 8-byte LDS read from
//     the pre-packed image of the cell it falls in (atlas, composite or scratch);
//   * a pass with more composited cells than the scratch holds falls back to
//     per-row 12 + 12-byte stores (bit-identical, 16 instead of 12 stores).
// The main loop of a rendering wave issues NO global loads (loads and stores
// share vmcnt and the per-CU memory pipe is in-order); the feeders' loads are
// few (a record is 6 KB against the 121-372 KB of pixels it turns into).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

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

// Cache policy of the observation stores (gfx950 sc0 / sc1 / nt bits), per
// instantiation (kNt).  The FUSED launch stores its pixels non-temporal: its
// feeders re-read the world records the previous launch wrote back (25 MB for
// 4096 clean_up worlds) while 495 MB of pixels stream out, and with plain stores
// those reads go to HBM in the middle of the write stream — where they cost far
// more than their bytes: the launch runs as fast with `nt` stores as it does with
// the record loads of batches >= 1 removed altogether (ablation, same box:
// 127.3 us plain, 118.2 us nt, 116.5 us without those loads; territory 411 /
// 374 / 356; sc1 and sc0 sc1, which drop the line from L2, are slower than
// plain; touching the records' cache lines at the start of the launch, while HBM
// idles, changes nothing with nt and costs 2 us; with the caches flushed between
// steps (`bench.py --cold`: 1 GiB streamed through) the launch takes 3 % longer,
// 5 % for territory; profiles/r03_store_policy.md).  The draw-only launch reads each record
// once, before its stores: plain stores are fastest there (round 1: nt +3 %).
template <bool kStep> constexpr bool nt_stores() { return kStep; }

namespace {

constexpr int kMaxLayers = 12;
constexpr int kSpriteStride = 272;  // 8*8*4 B + 16 B pad: spreads images over LDS banks
constexpr int kHeadBytes = 64;      // WorldTail head: ax[16], ay[16], aori[16], aalive[16]
// waves per workgroup: 16, i.e. 128 VGPRs a wave.  The step functions take
// 112-122 next to the renderer once the lane id is re-read per world (see the
// feeder loop); the matrix level's wants 135 and runs with 1-4 of them spilled
// (8-20 B of scratch per lane, touched on the rare interaction path): measured,
// prisoners_dilemma arena 365 us with 12-wave workgroups, 332 us with 16
// (profiles/r03_matrix_waves.md)
constexpr int kDrawThreads = 1024, kMatrixThreads = 1024;
constexpr int kMaxBatch = 8;        // worlds per batch
constexpr int kMaxSlots = 16;       // record slots of the ring (NB * B)
constexpr int kClaimRing = 32;      // claimed batches remembered (> NB + the claim distance)
constexpr int kMaxChains = 8;       // claim chains (= feeders / gcd(feeders, B))
constexpr uint32_t kNoBatch = 0xffffffffu;
constexpr int kStockHead = 1;       // FramePlan::head of the product

enum { FLAG_OPAQUE = 1, FLAG_PARTIAL = 2 };

// LDS image of a workgroup.  [0, world) is DevTables::render_blob verbatim.
struct FrameLds {
  int atlas, sinfo, rinfo, slot, stab, pairs, oobimg, world;   // the blob
  int step_tables;   // stepk tables (sinfo / spawn)
  int records;       // [NB][B] world records (world_stride each): the ring of resident batches
  int step_scratch;  // [feeders] stepk::Scratch + marks + substrate extra
  int recs, ovlist, offtab, ctrl, scratch, total;
};

// Pipeline state of a workgroup (LDS).
struct Ctrl {
  uint32_t next_ticket[2];           // (batch, pass) tickets per view, handed out in order
  uint32_t table_waves;              // feeder waves that have copied their share of the step tables
  uint32_t blob_waves;               // renderer waves that have copied their share of the blob
  uint32_t chain_end[kMaxChains];    // first batch of claim chain c (k % chains == c) that does not exist
  uint32_t done[kMaxSlots];          // passes completed in each ring buffer, ever (both views)
  uint32_t slot_batch[kMaxSlots];    // 1 + batch whose world sits in ring slot (buffer * B + position)
  uint32_t claim_tag[kClaimRing];    // 1 + batch whose first world is claim_w[same index]
  uint32_t claim_w[kClaimRing];      // first world of that batch, kNoBatch = the pool was empty
};

__host__ __device__ inline FrameLds frame_lds_layout(const DevTables& t, int slots, int feeders,
                                                     int nwaves, int slot_scratch_bytes) {
  FrameLds r;
  int off = 0;
  r.atlas = off; off += t.n_images * kSpriteStride;
  r.sinfo = off; off += 256 * 2;                                    // u16 per state
  r.rinfo = off; off += (((t.P + 1) * t.nsprites * 2) + 15) & ~15;  // u16 per (viewer, sprite)
  r.slot = off; off += ((t.nsprites * 4 * 2) + 15) & ~15;           // u16 per (sprite, facing)
  r.stab = off; off += 4 * 256 * 2;                                 // u16 per (facing, state)
  r.pairs = off; off += kPairSlots * 4;                             // composite cache
  r.oobimg = off; off += (((t.P + 1) * 2) + 15) & ~15;              // u16 per viewer: its OutOfBounds image
  r.world = off;
  r.step_tables = off; off += stepk::tables_bytes(t);
  r.records = off; off += slots * t.world_stride;
  r.step_scratch = off; off += feeders * slot_scratch_bytes;
  r.recs = off; off += nwaves * 64 * 16;                                 // per-wave draw lists
  r.ovlist = off; off += nwaves * 64;                                    // per-wave list of cells with overlays
  r.offtab = off; off += 2 * 64 * 4;                                     // per view
  r.ctrl = off; off += (int)sizeof(Ctrl);
  r.scratch = off; off += nwaves * t.scratch_cells * 256;                // per-wave composited images
  r.total = off;
  return r;
}

// What a launch needs of its plan, worked out on the host (frame_consts) and handed
// over as one argument struct.  Round 4, second session: a stamp at the kernel's
// very first instruction showed 4.4 us between it and the end of the prologue's barriers
// — before a single table was requested — spent on scalar housekeeping: thirty dependent
// s_load round trips into argument structs 0.9 KB long, gridDim / blockDim (the dispatch
// packet: two more lines), and a dozen integer divisions (batches, tickets per batch,
// strips per pass, the claim chains' gcd loop: ~45 instructions each on this ISA).  None
// of it depends on anything but the plan.
struct FrameConsts {
  FramePlan p;
  FrameLds lo;
  int32_t N, nbt;             // worlds, batches of the launch
  int32_t chains;             // claim chains A = F / gcd(F, B)
  int32_t pool_first;         // first pooled batch id (= groups * ks)
  int32_t b_mod_f;            // B % F (a batch's first slot, mod F, from its predecessor's)
  int32_t tables_vec, record_vec;   // 16-byte vectors of the step tables / of a record
  uint32_t first_k_nibbles[2];      // feeder f's first ring slot is slot f: nibble f = its batch
  // per view: [0] per-agent RGB, [1] WORLD.RGB
  int32_t row_cells[2], strip_rows[2], R[2], strips_per_world[2];
  uint32_t npb[2], magic_rows[2], magic_spw[2], magic_npb[2];
  uint32_t magic_p, npb_all, magic_nb;
  // the render planes that can show anything, bottom -> top (DevTables::vis_layers): how many,
  // each one's byte offset in a record (plane * H * W, two u16 per word), which hold avatar states
  int32_t nvis;
  uint32_t plane_off[6], av_planes;
};

// out = (src*a + dst*(255-a) + 127) / 255 per channel (A7); x/255 computed as
// (x + 1 + (x >> 8)) >> 8, exact for x < 65535 (max here 65152).
__device__ inline uint32_t blend_partial(uint32_t dst, uint32_t src) {
  // Branch-free: the formula is exact at a == 0 (-> dst) and a == 255 (-> src).
  // R and B are blended together in the two 16-bit halves of one register
  // (each field <= 255*255 + 127 + 255 < 2^16, so no carry crosses fields).
  const uint32_t a = src >> 24, ia = 255u - a;
  uint32_t rb = __umul24(src & 0xff00ffu, a) + __umul24(dst & 0xff00ffu, ia) + 0x7f007fu;
  rb = ((rb + 0x010001u + ((rb >> 8) & 0xff00ffu)) >> 8) & 0xff00ffu;
  uint32_t g = __umul24((src >> 8) & 255u, a) + __umul24((dst >> 8) & 255u, ia) + 127u;
  g = (g + 1u + (g >> 8)) >> 8;
  return rb | (g << 8);
}

// 24 bytes of one tile row at base + off (base wave-uniform, 8-byte aligned).
// `sc1` (wave-uniform, FramePlan::store_sc1): system-coherent stores instead of the
// instantiation's policy — the line leaves the XCD's L2 at once; on an output buffer the
// memory side serves unevenly that is 10 % faster for commons_harvest and slower for
// territory (profiles/r03_buffer_placement.md), so it is a dimension of the plan
// mp_tune times, not a constant.
template <bool kNt>
__device__ inline void store_row(uint8_t* base, uint32_t off, uint4 lo4, uint2 hi2, bool sc1) {
  // Two 12-byte stores (the form hipcc picks for a plain 24-byte struct copy
  // in tools/ubench/store_bw2.hip, which reaches 5.5 TB/s; a 16+8 split is
  // misaligned for every other cell and measures 2.1 TB/s).  Nothing ever
  // waits on these stores, so no vmcnt bookkeeping is needed around the asm.
  typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
  const u32x3 lo = {lo4.x, lo4.y, lo4.z}, hi = {lo4.w, hi2.x, hi2.y};
  if (sc1)
    asm volatile("global_store_dwordx3 %0, %1, %3 sc1\n\t"
                 "global_store_dwordx3 %0, %2, %3 offset:12 sc1"
                 :: "v"(off), "v"(lo), "v"(hi), "s"(base) : "memory");
  else if (kNt)
    asm volatile("global_store_dwordx3 %0, %1, %3 nt\n\t"
                 "global_store_dwordx3 %0, %2, %3 offset:12 nt"
                 :: "v"(off), "v"(lo), "v"(hi), "s"(base) : "memory");
  else
    asm volatile("global_store_dwordx3 %0, %1, %3\n\t"
                 "global_store_dwordx3 %0, %2, %3 offset:12"
                 :: "v"(off), "v"(lo), "v"(hi), "s"(base) : "memory");
}

// 16 bytes at base + off (16-byte aligned), or one 8-byte half of them.
template <bool kNt>
__device__ inline void store_chunk(uint8_t* base, uint32_t off, uint2 a, uint2 b, bool sc1) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {a.x, a.y, b.x, b.y};
  if (sc1) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" :: "v"(off), "v"(v), "s"(base));
  else if (kNt) asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(off), "v"(v), "s"(base));
  else asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(off), "v"(v), "s"(base));
}
template <int kOfs, bool kNt>
__device__ inline void store_half(uint8_t* base, uint32_t off, uint2 v2, bool sc1) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 v = {v2.x, v2.y};
  if (kOfs == 0) {
    if (sc1) asm volatile("global_store_dwordx2 %0, %1, %2 sc1" :: "v"(off), "v"(v), "s"(base));
    else if (kNt) asm volatile("global_store_dwordx2 %0, %1, %2 nt" :: "v"(off), "v"(v), "s"(base));
    else asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(off), "v"(v), "s"(base));
  } else {
    if (sc1) asm volatile("global_store_dwordx2 %0, %1, %2 offset:8 sc1" :: "v"(off), "v"(v), "s"(base));
    else if (kNt) asm volatile("global_store_dwordx2 %0, %1, %2 offset:8 nt" :: "v"(off), "v"(v), "s"(base));
    else asm volatile("global_store_dwordx2 %0, %1, %2 offset:8" :: "v"(off), "v"(v), "s"(base));
  }
}

// 8 RGB pixels (0x00BBGGRR each) <-> 24 packed bytes.
__device__ inline void pack_row(const uint32_t* px, uint32_t* w) {
  w[0] = px[0] | (px[1] << 24);
  w[1] = (px[1] >> 8) | (px[2] << 16);
  w[2] = (px[2] >> 16) | (px[3] << 8);
  w[3] = px[4] | (px[5] << 24);
  w[4] = (px[5] >> 8) | (px[6] << 16);
  w[5] = (px[6] >> 16) | (px[7] << 8);
}
__device__ inline void unpack_row(const uint32_t* w, uint32_t* px) {
  px[0] = w[0] & 0xffffffu;
  px[1] = (w[0] >> 24) | ((w[1] & 0xffffu) << 8);
  px[2] = (w[1] >> 16) | ((w[2] & 0xffu) << 16);
  px[3] = w[2] >> 8;
  px[4] = w[3] & 0xffffffu;
  px[5] = (w[3] >> 24) | ((w[4] & 0xffffu) << 8);
  px[6] = (w[4] >> 16) | ((w[5] & 0xffu) << 16);
  px[7] = w[5] >> 8;
}

// Composite one sprite row (8 px) onto the row held in registers.
// The 32-bit words of a small POD of wave-uniform values: in scalar registers as of here,
// and no longer traceable to where they were loaded from.
template <class S>
__device__ inline void pin_scalars(S& s) {
  static_assert(sizeof(S) % 4 == 0, "a POD of 32-bit words");
  uint32_t w[sizeof(S) / 4];
  __builtin_memcpy(w, &s, sizeof(S));
#pragma unroll
  for (size_t i = 0; i < sizeof(S) / 4; ++i) asm volatile("" : "+s"(w[i]));
  __builtin_memcpy(&s, w, sizeof(S));
}

// ... read (and only read) here: the loads cannot be moved below this point.
template <class S>
__device__ inline void touch_scalars(const S& s) {
  if constexpr (sizeof(S) >= 4) {
    uint32_t w[sizeof(S) / 4];
    __builtin_memcpy(w, &s, sizeof(w));
#pragma unroll
    for (size_t i = 0; i < sizeof(S) / 4; ++i) asm volatile("" ::"s"(w[i]));
  }
}

// Keeps the 32-bit words of a small POD in registers as of here (see stepk::issued).
template <class S>
__device__ inline void pin_words(S& s) {
  if constexpr (sizeof(S) >= 4) {
    static_assert(sizeof(S) % 4 == 0, "a POD of 32-bit words");
    uint32_t w[sizeof(S) / 4];
    __builtin_memcpy(w, &s, sizeof(S));
#pragma unroll
    for (size_t i = 0; i < sizeof(S) / 4; ++i) asm volatile("" : "+v"(w[i]));
    __builtin_memcpy(&s, w, sizeof(S));
  }
}

template <int kMode>  // 1: binary alpha, 2: 8-bit blend
__device__ inline void blend_row(uint32_t* acc, const uint8_t* row) {
  const uint4* src = reinterpret_cast<const uint4*>(row);
  const uint4 a = src[0], b = src[1];
  const uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (kMode == 1) acc[j] = (s[j] >> 24) ? (s[j] & 0xffffffu) : acc[j];
    else acc[j] = blend_partial(acc[j], s[j]);
  }
}

__device__ inline uint32_t fast_div(uint32_t n, uint32_t d, float rcp) {
  uint32_t q = (uint32_t)((float)n * rcp);
  if (q * d > n) --q;
  else if ((q + 1) * d <= n) ++q;
  return q;
}

// n / d in one multiply, for the divisions a pass repeats: magic = 2^32 / d + 1
// (0 for d == 1) is exact while n * d < 2^32 — strips and images of one batch
// are thousands at most.
__host__ __device__ inline uint32_t div_magic(uint32_t d) {
  return d == 1u ? 0u : (uint32_t)(0x100000000ull / d) + 1u;
}
__device__ inline uint32_t magic_div(uint32_t n, uint32_t magic) {
  return magic == 0u ? n : __umulhi(n, magic);
}

// Draw list of one output cell: byte offset of the opaque base image in the LDS
// atlas (image 0 = black when there is none; kSkipCopy set when phase 2a must
// leave the cell alone) + up to 8 overlay entries of 12 bits (flags << 10 |
// image), bottom -> top from bit 0.
struct CellRec { uint32_t base, ov0, ov1, ov2; };
constexpr uint32_t kSkipCopy = 0x80000000u;  // in CellRec::base: not a plain single-image cell
constexpr uint32_t kDeadCell = 0x40000000u;  // ... because it is beyond the last strip
constexpr uint32_t kAvatarBit = 0x8000u;


}  // namespace
namespace stepk {
struct NoTables {};   // render-only instantiation: the feeders only load records
struct NoSites {};
__device__ inline NoSites load_sites(const NoTables&, int) { return NoSites(); }
}  // namespace stepk
namespace {
using stepk::NoSites;
using stepk::NoTables;
template <class Tables> constexpr int max_threads() {
  return std::is_same<Tables, MatrixTables>::value ? kMatrixThreads : kDrawThreads;
}

// Global -> LDS without registers (global_load_lds_*, gfx950): lane l's 16 (4) bytes
// land at `lds` + 16 l (4 l); inactive lanes write nothing.  M0 carries the LDS byte
// address and is put back.  The compiler does not see the LDS write — and must not: it
// answers the builtin form with an s_waitcnt vmcnt(0) in front of the first LDS read of
// EVERY iteration of a loop that follows, stores in flight included — so whoever reads
// what was requested waits with dma_wait() first.
__device__ inline uint32_t lds_byte_address(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)p;
}
__device__ inline void dma_b128(const void* g, const void* lds) {
  const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_byte_address(lds));
  uint32_t m0_was;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_was) : "v"(g), "s"(a) : "memory");
}
__device__ inline void dma_b32(const void* g, const void* lds) {
  const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_byte_address(lds));
  uint32_t m0_was;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_was) : "v"(g), "s"(a) : "memory");
}
__device__ inline void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ inline uint32_t lds_acquire(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Every wait of the pipeline is bounded in WALL time (the 100 MHz constant clock,
// not a poll count: a profiler or sanitizer may slow the polling loop down by
// orders of magnitude): a wave that has waited 2 s gives up,
// records where (DevTables::fault: {site, workgroup, wave, batch, seen, wanted})
// and leaves; the host reports it at its next synchronising call instead of
// hanging on a kernel that will never finish.
// (developer build -DMP_FRAME_TRACE: every wave of workgroup 0 also leaves its
// last pipeline stage in fault[16 + wave]; the fault words live in host memory,
// so they can be read while a kernel is stuck)
#if defined(MP_FRAME_TIMELINE)
// developer build: every wave of workgroups 0, 1, 128 and 255 logs (stage |
// value << 8, wall clock) pairs behind the fault words (tools/gpu_timeline.py)
constexpr int kTimelineEvents = 64;   // per wave
#define FRAME_STAGE(code, value)                                                        \
  do {                                                                                  \
    const int tl_wg = blockIdx.x == 0 ? 0 : blockIdx.x == 1 ? 1 : blockIdx.x == 128 ? 2  \
                      : (int)blockIdx.x == K.p.groups - 1 ? 3 : -1;                        \
    if (tl_wg >= 0 && lane == 0 && tl_n < kTimelineEvents) {                            \
      uint32_t* tl = t.fault + 64 + ((tl_wg * 16 + wave) * kTimelineEvents + tl_n) * 2; \
      tl[0] = (uint32_t)(code) | ((uint32_t)(value) << 8);                              \
      tl[1] = (uint32_t)wall_clock64();                                                 \
    }                                                                                   \
    ++tl_n;                                                                             \
  } while (0)
#elif defined(MP_FRAME_TRACE)
#ifndef MP_TRACE_MASK
#define MP_TRACE_MASK 0xffffu
#endif
#define FRAME_STAGE(code, value)                                                        \
  do {                                                                                  \
    if (((MP_TRACE_MASK >> (code)) & 1u) && blockIdx.x == 0 && lane == 0) {             \
      __hip_atomic_store(&t.fault[16 + wave], (uint32_t)(code) | ((uint32_t)(value) << 8), \
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);                  \
    }                                                                                   \
  } while (0)
#else
#define FRAME_STAGE(code, value)
#endif
constexpr uint64_t kMaxWaitTicks = 200000000ull;   // 2 s of wall_clock64()
enum { FAULT_BUFFER_FREE = 1, FAULT_BATCH_READY = 2, FAULT_PROLOGUE = 3, FAULT_CLAIM = 4 };
// true once a wait that started at its first call (t0 == 0) has lasted too long;
// the clock is read every 256th poll only
__device__ inline bool waited_too_long(uint32_t polls, uint64_t& t0) {
  if ((polls & 255u) != 255u) return false;
  const uint64_t now = wall_clock64();
  if (t0 == 0) { t0 = now; return false; }
  return now - t0 > kMaxWaitTicks;
}
__device__ inline void report_stall(const DevTables& t, int lane, uint32_t site, uint32_t wave,
                                    uint32_t batch, uint32_t seen, uint32_t wanted) {
  // (every lane tries: exactly one wins the word, no lane predicate to merge
  // with the loop's own — see the ticket loop)
  (void)lane;
  if (atomicCAS(&t.fault[0], 0u, site) == 0u) {
    t.fault[1] = blockIdx.x; t.fault[2] = wave; t.fault[3] = batch;
    t.fault[4] = seen; t.fault[5] = wanted;
  }
}

// kViews: 0 = the per-agent view (out_a), 1 = WORLD.RGB (out_w), 2 = both in one launch
// (the last plan.world_waves renderer waves draw WORLD.RGB, the others the per-agent view,
// from the same LDS-resident records)
template <class Tables, class Sites, int kViews>
__global__ __launch_bounds__(max_threads<Tables>()) void k_frame(DevTables t, Tables c,
                                                       stepk::StepArgs args,
                                                       uint8_t* __restrict__ out_a,
                                                       uint8_t* __restrict__ out_w,
                                                       FrameConsts K) {
  constexpr bool kStep = !std::is_same<Tables, NoTables>::value;
  constexpr bool kNt = nt_stores<kStep>();
#if defined(MP_FRAME_TIMELINE)
  const uint64_t tl_entry = wall_clock64();
#endif
  {
    // Warm the scalar cache with the kernel's arguments (~0.9 KB by value: the
    // table structs).  The compiler fetches them where they are first needed, in
    // dependent batches: seven s_load / s_waitcnt round trips in a row in front of
    // the feeders' first step, 2.4 us on the critical path of the launch when each
    // one misses.  One dword per 64-byte line, all in flight at once, here.
    constexpr int kArgBytes = (int)(sizeof(DevTables) + sizeof(Tables) + sizeof(stepk::StepArgs) +
                                    2 * sizeof(uint8_t*) + sizeof(FrameConsts));
    typedef const uint32_t __attribute__((address_space(4))) KernargWord;
    KernargWord* ka = (KernargWord*)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t warm = 0;
#pragma unroll
    for (int i = 0; i < kArgBytes / 4; i += 16) warm ^= ka[i];
    asm volatile("" ::"s"(warm));
  }
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const FramePlan plan = K.p;
  const FrameLds lo = K.lo;
  const struct { int32_t N, nbt, chains, pool_first, b_mod_f, tables_vec, record_vec;
                 uint32_t first_k0, first_k1, magic_p, npb_all; } kc = {
      K.N, K.nbt, K.chains, K.pool_first, K.b_mod_f, K.tables_vec, K.record_vec,
      K.first_k_nibbles[0], K.first_k_nibbles[1], K.magic_p, K.npb_all};
  const int kWaves = plan.nwaves;
  const int B = plan.B, NB = plan.NB;
  const int F = plan.feeders;
  const bool sc1 = __builtin_amdgcn_readfirstlane(plan.store_sc1) != 0;
  const int pace = __builtin_amdgcn_readfirstlane(plan.pace);
  const int tid = threadIdx.x;
  const int HW = t.H * t.W, L = t.L, P = t.P, W = t.W, H = t.H;
  uint8_t* atlas = smem + lo.atlas;
  uint16_t* sinfo = reinterpret_cast<uint16_t*>(smem + lo.sinfo);  // sprite | (player+1) << 8
  uint16_t* rinfo = reinterpret_cast<uint16_t*>(smem + lo.rinfo);  // remapped sprite | flags << 8
  uint16_t* slot = reinterpret_cast<uint16_t*>(smem + lo.slot);    // atlas image of (sprite, facing)
  uint16_t* stab = reinterpret_cast<uint16_t*>(smem + lo.stab);    // entry of (facing, state)
  uint32_t* pairs = reinterpret_cast<uint32_t*>(smem + lo.pairs);
  const uint16_t* oobimg = reinterpret_cast<const uint16_t*>(smem + lo.oobimg);
  const int wstride = t.world_stride;                              // a whole record per world
  Ctrl* ctrl = reinterpret_cast<Ctrl*>(smem + lo.ctrl);

  // (read through the scalar unit: the feeder / renderer branch below must be
  // provably wave-uniform, or both paths' registers stay live across each other)
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_render_waves = kWaves - F;
  // the view this wave draws (feeders: neither)
  const bool wv = kViews == 1 || (kViews == 2 && wave >= n_render_waves - plan.world_waves);
  const struct { int32_t VW, VH, row_cells, strip_rows, R, strips_per_world;
                 uint32_t npb, magic_rows, magic_spw, magic_npb;
                 int32_t nvis; uint32_t plane_off[6], av_planes; } kv = {
      K.row_cells[0], K.strip_rows[0], wv ? K.row_cells[1] : K.row_cells[0],
      wv ? K.strip_rows[1] : K.strip_rows[0], wv ? K.R[1] : K.R[0],
      wv ? K.strips_per_world[1] : K.strips_per_world[0], wv ? K.npb[1] : K.npb[0],
      wv ? K.magic_rows[1] : K.magic_rows[0], wv ? K.magic_spw[1] : K.magic_spw[0],
      wv ? K.magic_npb[1] : K.magic_npb[0],
      K.nvis, {K.plane_off[0], K.plane_off[1], K.plane_off[2], K.plane_off[3], K.plane_off[4],
               K.plane_off[5]}, K.av_planes};
  const int VW = kv.VW, VH = kv.VH;
  const int row_cells = kv.row_cells;
  const int strip_rows = kv.strip_rows;   // strips per image
  const uint32_t row_bytes = (uint32_t)row_cells * 24u;
  const int R = kv.R;                     // strips per wave pass (64 / row_cells)
  const int sr = (int)fast_div((uint32_t)lane, (uint32_t)row_cells, 1.0f / (float)row_cells);
  const uint32_t cx = (uint32_t)(lane - sr * row_cells);
  uint32_t* offtab = reinterpret_cast<uint32_t*>(smem + lo.offtab) + (wv ? 64 : 0);

#if defined(MP_FRAME_TIMELINE)
  int tl_n = 0;
#endif
  FRAME_STAGE(1, 0);
#if defined(MP_FRAME_TIMELINE)
  FRAME_STAGE(19, (uint32_t)(wall_clock64() - tl_entry));   // 10 ns ticks since the first instruction
#endif
  // ---- which worlds.  The launch's worlds are cut into batches of B (batch id b =
  // worlds [b * B, b * B + B)); this workgroup OWNS the ids [g * ks, (g + 1) * ks) — a
  // contiguous range, walked first — and then claims ids beyond groups * ks one at a
  // time from a device-wide counter until the pool is empty.  The 8 XCDs do not get
  // equal shares of a saturated memory system (their workgroups finish an even split
  // 58 ... 97 us after the start, in IOD pairs, differently for every output buffer:
  // profiles/r04_write_fronts.md), so an even split leaves the fast ones idle at the
  // end; the pool is what they take instead.
  const int N = kc.N;
  const int ks = plan.ks;
  const int nbt = kc.nbt;                                 // batches in the launch
  const int pool_first = kc.pool_first;                   // first pooled batch id
  // Which worlds a workgroup OWNS: batch k starts at world w_first + k * kstep while that is
  // < w_end.  Stock: its own contiguous range (kstep = B).  FramePlan::team (single-world
  // batches only; round 6, second form): the workgroups of XCD x (workgroup g runs on XCD
  // g % 8: observed, used for speed only) are a team that shares one contiguous range of
  // worlds — as long as its members' ranges together — and member j = g / 8 of its m takes the
  // team's worlds j, j + m, j + 2 m ...: every XCD writes ONE compact front (its 32 workgroups
  // draw 32 neighbouring worlds at a time) instead of 32 fronts two megabytes apart — the order
  // the bare store loop takes 8 - 13 us faster on the buffers the memory side serves unevenly and
  // no slower on the others (profiles/r04_write_fronts.md).  With the old resolve the renderers,
  // not the memory side, paced the launch and the order bought nothing (profiles/r06_team_deal.md);
  // with the new one the launch IS its store loop.
  int w_first = (int)blockIdx.x * ks * B, kstep = B, w_end = N;
  if (__builtin_amdgcn_readfirstlane(plan.team) != 0) {
    const int G = plan.groups, x = (int)blockIdx.x & 7, q = G >> 3, r = G & 7;
    const int m = q + (x < r ? 1 : 0);                    // members of this team
    const int start = (q * x + (x < r ? x : r)) * ks;     // the teams before it, whole (B == 1)
    w_first = start + ((int)blockIdx.x >> 3);
    kstep = m;
    w_end = start + m * ks;
    if (w_end > N) w_end = N;
  }
  // claim chains: the feeder that owns slot 0 of batch k owns slot 0 of batch k + A too
  // (A = F / gcd(F, B)); when it starts batch k it claims batch k + A, so a claim's trip
  // to the counter overlaps a whole step.  Chain c = the batches k % A == c.
  const int A = kc.chains;
  const int strips_per_world = kv.strips_per_world;
  const uint32_t npb = kv.npb;   // this view's tickets per batch
  // passes of a batch over all views (what frees its buffer)
  const uint32_t npb_all = kc.npb_all;

  // ---- prologue: what never changes, into LDS (once per workgroup).  The two
  // roles part at once: the feeders need the step tables (1.5 KB) and nothing of
  // the render blob (50+ KB), and the first batch's steps are the critical path of
  // the launch — so the feeders copy the tables themselves and start stepping
  // after ~2 us, while the renderers copy the blob and build their key tables
  // (the one workgroup barrier left only orders the zeroing of the pipeline state;
  // each role then meets at its own LDS arrival counter)
  if (tid < (int)(sizeof(Ctrl) / 4)) reinterpret_cast<uint32_t*>(ctrl)[tid] = 0u;
  __syncthreads();
  if (tid < kMaxChains) ctrl->chain_end[tid] = plan.pool > 0 ? kNoBatch : (uint32_t)ks;
  // the counter the NEXT frame launch will claim from starts at zero — whatever this
  // launch's own plan: launches with and without a pool alternate (a draw-only
  // mp_observe between two steps, mp_tune's candidates)
  if (blockIdx.x == 0 && tid == 0)
    __hip_atomic_store(&t.claim[plan.parity ^ 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  FRAME_STAGE(16, 0);
  Sites sites = Sites();
  const int head_mode = kStep ? __builtin_amdgcn_readfirstlane(plan.head) : 0;
  int pre_w = -1, pre_slot = -1;   // the world a feeder requested ahead (head & 1), its ring slot
  bool head_pending = false;       // ... and has not waited for yet
  auto arrive_and_wait = [&](uint32_t* counter, uint32_t want) -> bool {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicAdd(counter, 1u);
    uint64_t wait_t0 = 0;
    for (uint32_t polls = 0; lds_acquire(counter) < want; ++polls) {
      if (waited_too_long(polls, wait_t0)) {
        report_stall(t, lane, FAULT_PROLOGUE, (uint32_t)wave, 0u, lds_acquire(counter), want);
        return false;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    return true;
  };
  if (wave >= n_render_waves) {
    if (kStep && (head_mode & 1)) {
      __builtin_amdgcn_s_setprio(3);   // (already here: the feeders' prologue wins the issue slots)
      // Round 4, second session: NOTHING is waited for here.  The site lists are requested
      // (registers); the tables (this feeder's KiB chunks of them) and the record of the
      // first world this feeder will step go global -> LDS by DMA, its action ids into the
      // wave's unused draw-list area; the scratch is set up; and the loop is entered — the
      // values LICM hoists out of the step (~800 scalar / vector instructions in front of
      // the loop, 1.6 us) are computed while all of that is in flight.  The feeder waits,
      // files its tables and meets the others at its first world (`head_pending`).
      // Before: tables back at 2.8 us, hoisted values until 4.6, the first record requested
      // at 5.0 (profiles/r03_frame_timeline.md).
      const int f = wave - n_render_waves;
      // the first ring slot this feeder owns is slot f (F <= NB * B), in batch k = f / B
      // (the host's division) — if that batch is one this workgroup OWNS (arithmetic
      // index) and the world exists; a pooled first batch takes the old road
      {
        const int k = (int)(((f < 8 ? kc.first_k0 : kc.first_k1) >> (4 * (f & 7))) & 15u);
        const int sl = f - k * B;
        const int w = w_first + k * kstep + sl;
        if (k < ks && w < w_end) { pre_w = w; pre_slot = f; }
      }
      if (pre_w >= 0) {
        const uint4* rsrc = reinterpret_cast<const uint4*>(args.state + (size_t)pre_w * wstride) + lane;
        uint8_t* rdst = smem + lo.records + pre_slot * wstride;
        const int nvec = kc.record_vec;
        int j = 0;
        for (; j + 64 <= nvec; j += 64) dma_b128(rsrc + j, rdst + j * 16);
        if (j + lane < nvec) dma_b128(rsrc + j, rdst + j * 16);
        if (args.mode == STEP_MODE_STEP && lane < P)
          dma_b32(args.actions + (size_t)pre_w * P + lane, smem + lo.recs + wave * 64 * 16);
      }
      {
        const int tvec = kc.tables_vec;
        const uint4* tsrc = reinterpret_cast<const uint4*>(t.step_blob) + lane;
        for (int j = f * 64; j < tvec; j += F * 64)
          if (j + lane < tvec) dma_b128(tsrc + j, smem + lo.step_tables + j * 16);
      }
      FRAME_STAGE(17, 0);
      sites = stepk::load_sites(c, lane);
      FRAME_STAGE(18, 0);
      uint8_t* scratch0 = smem + lo.step_scratch + f * plan.slot_scratch;
      stepk::clear_marks(t, scratch0 + sizeof(stepk::Scratch), lane);
      stepk::wsync();
      stepk::init_extra(t, c, scratch0 + stepk::scratch_bytes(t), lane);
      head_pending = true;
      if (pre_w < 0) {   // no world to wait at: wait here
        dma_wait();
        pin_words(sites);
        head_pending = false;
        if (!arrive_and_wait(&ctrl->table_waves, (uint32_t)F)) return;
      }
    } else if (kStep) {
      // The older road (MpDevOptions.head = 1; a pooled first batch never comes here:
      // head_mode is per launch).  A feeder's set-up — its site lists (global, L2-resident),
      // its scratch's marks and extras (LDS) — does not need the tables: it runs while the
      // tables' loads are in flight instead of after the feeders have met (4 us of set-up
      // in a row before: tables 2.6, site lists 1.2, marks 0.7, extras 0.2).  The site lists
      // are pinned (stepk::issued): the compiler otherwise sinks their loads to the first
      // use, a round trip inside the first step.
      // (Tried and dropped in round 3, profiles/r03_frame_timeline.md: requesting the first
      // world's action ids and its record here as well, through registers — loads that go
      // to HBM next to the blob copy held the tables back with them.  The DMA head above
      // holds no register and waits for nothing before the first step.)
      sites = stepk::load_sites(c, lane);
      const int ftid = tid - n_render_waves * 64, fthreads = F * 64;
      const int tvec = stepk::tables_bytes(t) >> 4;   // <= 1.5 KB: at most two per thread
      const uint4* tsrc = reinterpret_cast<const uint4*>(t.step_blob);
      uint4 ta = {}, tb = {};
      if (ftid < tvec) ta = tsrc[ftid];
      if (ftid + fthreads < tvec) tb = tsrc[ftid + fthreads];
      stepk::issued(ta); stepk::issued(tb);
      uint8_t* scratch0 = smem + lo.step_scratch + (wave - n_render_waves) * plan.slot_scratch;
      stepk::clear_marks(t, scratch0 + sizeof(stepk::Scratch), lane);
      stepk::wsync();
      stepk::init_extra(t, c, scratch0 + stepk::scratch_bytes(t), lane);
      uint4* tdst = reinterpret_cast<uint4*>(smem + lo.step_tables);
      if (ftid < tvec) tdst[ftid] = ta;
      if (ftid + fthreads < tvec) tdst[ftid + fthreads] = tb;
      for (int i = ftid + 2 * fthreads; i < tvec; i += fthreads) tdst[i] = tsrc[i];   // (bigger tables)
      pin_words(sites);
      if (!arrive_and_wait(&ctrl->table_waves, (uint32_t)F)) return;
    }
  }
  // The renderers' share of the prologue.  (Tried, round 4: the keys below built BEFORE
  // the copy, the copy not before the feeders have their first data — either way the
  // renderers compete with the feeders' first instructions or are ready too late; no gain.)
  if (wave < n_render_waves) {
    const uint4* src = reinterpret_cast<const uint4*>(t.render_blob);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const int n = lo.world >> 4, nthr = n_render_waves * 64;
    // eight loads in flight per thread: the copy is latency-, not bandwidth-bound
    for (int i = tid; i < n; i += 8 * nthr) {
      uint4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src[min(i + k * nthr, n - 1)];
#pragma unroll
      for (int k = 0; k < 8; ++k) stepk::issued(v[k]);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i + k * nthr < n) dst[i + k * nthr] = v[k];
    }
    // (one wave per view writes that view's table; with a single view, wave 0)
    if (wave == 0 || (kViews == 2 && wave == n_render_waves - plan.world_waves))
      offtab[lane] = (uint32_t)sr * 8u * row_bytes + cx * 24u;
    FRAME_STAGE(2, ks);
    if (!arrive_and_wait(&ctrl->blob_waves, (uint32_t)n_render_waves)) return;
  }
  FRAME_STAGE(3, npb);

  // First world of this workgroup's k-th batch; -1 = there is no such batch (the
  // pool was empty when its turn came; `stalled` = gave up waiting for the claim).
  // Owned batches are arithmetic; a pooled one is known once its claim has come back.
  auto batch_first_world = [&](int k, bool& stalled) -> int {
    if (k < ks) {
      const int w0 = w_first + k * kstep;
      return w0 < w_end ? w0 : -1;     // (the last workgroup's / team's range may run past the end)
    }
    const int ring = k % kClaimRing, chain = k % A;
    uint64_t wait_t0 = 0;
    for (uint32_t polls = 0;; ++polls) {
      if (lds_acquire(&ctrl->claim_tag[ring]) == (uint32_t)(k + 1)) {
        const uint32_t w0 = ctrl->claim_w[ring];
        return w0 == kNoBatch ? -1 : (int)w0;
      }
      if (lds_acquire(&ctrl->chain_end[chain]) <= (uint32_t)k) return -1;
      if (waited_too_long(polls, wait_t0)) {
        report_stall(t, lane, FAULT_CLAIM, (uint32_t)wave, (uint32_t)k,
                     lds_acquire(&ctrl->claim_tag[ring]), (uint32_t)(k + 1));
        stalled = true;
        return -1;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  };
  // no batch at or after k exists in any chain
  auto all_chains_ended = [&](int k) -> bool {
    uint32_t last = 0;
    for (int ch = 0; ch < A; ++ch) {
      const uint32_t e = lds_acquire(&ctrl->chain_end[ch]);
      last = e > last ? e : last;
    }
    return last <= (uint32_t)k;
  };

  // ---- feeders: the last F waves.  The NB buffers are a ring of NB * B slots
  // (slot r = buffer * B + position); feeder f brings the worlds of the ring
  // slots r = f, f + F, ... into LDS (and steps them): with F <= B every feeder
  // works on every batch, with F = NB * B a feeder owns one slot.  They run ahead
  // as far as the buffers allow.
  const int role_wave = wave;
  if (role_wave >= kWaves - F) {
    const int f = wave - (kWaves - F);
    // The feeders' copies of what their loop reads, in scalar registers as of here: a value
    // the compiler can trace to the argument segment is not kept (or spilled to a VGPR
    // lane) under register pressure but RE-LOADED where it is used — s_load + s_waitcnt
    // lgkmcnt(0), LDS reads in flight or not: 200 - 240 scalar loads in a stepping kernel
    // instead of 57, most of them in this loop and in the step it calls.  (The renderers
    // keep the traceable values: pinned for them too, their passes carry twice the
    // v_readlane traffic and WORLD.RGB is 5 % slower.)
    struct { int32_t B, NB, F, ks, N, nbt, A, pool_first, b_mod_f, wstride, pool, parity,
                     late_prio, records, step_tables, recs, w_first, kstep, w_end;
             uint32_t npb_all; } fc = {
        B, NB, F, ks, N, nbt, A, pool_first, kc.b_mod_f, wstride, plan.pool, plan.parity,
        plan.late_prio, lo.records, lo.step_tables, lo.recs, w_first, kstep, w_end, npb_all};
    pin_scalars(fc);
    auto batch_first_world = [&](int k, bool& stalled) -> int {   // (as the renderers' below)
      if (k < fc.ks) {
        const int w0 = fc.w_first + k * fc.kstep;
        return w0 < fc.w_end ? w0 : -1;
      }
      const int ring = k % kClaimRing, chain = k % fc.A;
      uint64_t wait_t0 = 0;
      for (uint32_t polls = 0;; ++polls) {
        if (lds_acquire(&ctrl->claim_tag[ring]) == (uint32_t)(k + 1)) {
          const uint32_t w0 = ctrl->claim_w[ring];
          return w0 == kNoBatch ? -1 : (int)w0;
        }
        if (lds_acquire(&ctrl->chain_end[chain]) <= (uint32_t)k) return -1;
        if (waited_too_long(polls, wait_t0)) {
          report_stall(t, lane, FAULT_CLAIM, (uint32_t)wave, (uint32_t)k,
                       lds_acquire(&ctrl->claim_tag[ring]), (uint32_t)(k + 1));
          stalled = true;
          return -1;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    };
    auto all_chains_ended = [&](int k) -> bool {
      uint32_t last = 0;
      for (int ch = 0; ch < fc.A; ++ch) {
        const uint32_t e = lds_acquire(&ctrl->chain_end[ch]);
        last = e > last ? e : last;
      }
      return last <= (uint32_t)k;
    };
    // A step is a chain of dependent instructions: whenever its next one is ready
    // it should issue ahead of the renderers' (which have plenty of independent
    // work per wave and give up next to nothing)
    __builtin_amdgcn_s_setprio(3);
    uint8_t* my_scratch = smem + lo.step_scratch + f * plan.slot_scratch;
    FRAME_STAGE(10, 0);
    bool first_world = true;
    // (no division in here: kb = k % NB, gen = k / NB, r0 = kb * B, m = r0 % F are carried)
    int kb = 0, gen = 0, r0 = 0, m = 0;
    auto next_batch = [&]() {
      ++kb; r0 += fc.B; m += fc.b_mod_f;
      if (m >= fc.F) m -= fc.F;
      if (kb == fc.NB) { kb = 0; r0 = 0; m = 0; ++gen; }
    };
    for (int k = 0;; ++k, next_batch()) {
      // (does this feeder own a slot of batch k at all?  It owns the ring slots r = f
      // (mod F), every feeder NB * B / F of them: in this batch sl = mine, mine + F, ...)
      const int mine = f >= m ? f - m : f - m + fc.F;
      if (mine >= fc.B) continue;
      FRAME_STAGE(4, k);
      bool stalled = false;
      const int w0 = batch_first_world(k, stalled);
      if (stalled) return;
      if (w0 < 0) {   // this chain's pool ran dry; batches of other chains may still come
        if (all_chains_ended(k)) break;
        __builtin_amdgcn_s_sleep(8);
        continue;
      }
      int nw = fc.N - w0;
      if (nw > fc.B) nw = fc.B;
      uint64_t wait_t0 = 0;
      // (ring buffer kb may take batch k once every pass of batch k - NB is done)
      for (uint32_t polls = 0;
           gen > 0 && lds_acquire(&ctrl->done[kb]) < (uint32_t)gen * fc.npb_all; ++polls) {
        if (waited_too_long(polls, wait_t0)) {
          report_stall(t, lane, FAULT_BUFFER_FREE, (uint32_t)wave, (uint32_t)k,
                       lds_acquire(&ctrl->done[kb]), (uint32_t)gen * fc.npb_all);
          return;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      for (int sl = mine; sl < fc.B; sl += fc.F) {
        FRAME_STAGE(5, sl);
        // the owner of a batch's first slot claims this chain's next batch: the
        // atomic goes out ahead of the record's loads and has come back, memory
        // returning in order, when they have
        const bool claims = sl == 0 && k + fc.A >= fc.ks && fc.pool > 0;
        uint32_t claimed = 0;
        if (claims && lane == 0)
          claimed = __hip_atomic_fetch_add(&t.claim[fc.parity], 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
        const int w = w0 + sl;
        if (sl < nw) {
          uint8_t* rec = smem + fc.records + (r0 + sl) * fc.wstride;
          if constexpr (kStep) {
            // the lane id is re-read per world: everything a step derives from it
            // (beam footprint cell, draw indices, masks) would otherwise be
            // hoisted out of the two loops and held in registers across them —
            // 150+ VGPRs for a function that needs 60 when it runs once
            int lane_w = lane;
            asm volatile("" : "+v"(lane_w));
            stepk::World wd = stepk::make_world(t, rec, smem + fc.step_tables, my_scratch,
                                                args.state, w, lane_w);
            wd.publish = &ctrl->slot_batch[r0 + sl];   // (finish(): as soon as the record is final)
            wd.publish_value = (uint32_t)(k + 1);
            wd.next_orders = args.next_orders;
            // (head & 1) this feeder's first world: what the prologue requested is waited
            // for HERE — tables, record, action ids — and the feeders meet
            bool have_rec = false;
            if (head_pending) {
              head_pending = false;
              dma_wait();   // (everything requested in the prologue, the site lists included)
              have_rec = w == pre_w && r0 + sl == pre_slot;
              if (!arrive_and_wait(&ctrl->table_waves, (uint32_t)fc.F)) return;
              FRAME_STAGE(11, sl);
            }
            int act_id;
            if (have_rec && args.mode == STEP_MODE_STEP)
              act_id = lane_w < P ? reinterpret_cast<const int*>(smem + fc.recs + wave * 64 * 16)[lane_w] : 0;
            else
              act_id = stepk::fetch_action_id(t, args.actions, args.mode, w, lane_w);
            if (!have_rec) stepk::load_record(t, rec, wd.gw, lane_w);
            FRAME_STAGE(12, sl);
            if (claims) {
              if (lane == 0) {
                const uint32_t id = (uint32_t)fc.pool_first + claimed;
                const int kn = k + fc.A, ring = kn % kClaimRing;
                const bool have = id < (uint32_t)fc.nbt;
                ctrl->claim_w[ring] = have ? id * (uint32_t)fc.B : kNoBatch;
                if (!have)
                  __hip_atomic_store(&ctrl->chain_end[kn % fc.A], (uint32_t)kn, __ATOMIC_RELEASE,
                                     __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&ctrl->claim_tag[ring], (uint32_t)(kn + 1), __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
            stepk::begin_step(wd.sc, lane_w);
            stepk::wsync();
            const stepk::Action act = stepk::lookup_action(t, wd, act_id, args.mode);
            stepk::step_world(t, c, sites, wd, act, args);
          } else {
            stepk::load_record(t, rec, args.state + (size_t)w * fc.wstride, lane);
          }
        }
        if (claims && (!kStep || sl >= nw)) {
          if (lane == 0) {
            const uint32_t id = (uint32_t)fc.pool_first + claimed;
            const int kn = k + fc.A, ring = kn % kClaimRing;
            const bool have = id < (uint32_t)fc.nbt;
            ctrl->claim_w[ring] = have ? id * (uint32_t)fc.B : kNoBatch;
            if (!have)
              __hip_atomic_store(&ctrl->chain_end[kn % fc.A], (uint32_t)kn, __ATOMIC_RELEASE,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&ctrl->claim_tag[ring], (uint32_t)(kn + 1), __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        // publish: the record's LDS writes are ordered before the flag
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0)
          __hip_atomic_store(&ctrl->slot_batch[r0 + sl], (uint32_t)(k + 1), __ATOMIC_RELEASE,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
        FRAME_STAGE(6, sl);
        // Only a feeder's first world is on the critical path (nothing can be drawn
        // before the first batch); every later one has a whole batch's drawing
        // time, so from then on the feeders stop taking issue slots from the
        // renderers — unless a step is so long (territory: 20+ us alone) that it
        // would then miss its turn (plan.late_prio; profiles/r03_store_policy.md)
        if (first_world) {
          first_world = false;
          switch (fc.late_prio) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: break;
          }
        }
      }
    }
    FRAME_STAGE(15, 0);
    return;
  }

  // ---- renderers
  uint8_t* __restrict__ out = wv ? out_w : out_a;
  CellRec* recs = reinterpret_cast<CellRec*>(smem + lo.recs) + wave * 64;
  uint8_t* ovlist = smem + lo.ovlist + wave * 64;
  // (a copy of its own: the kernel arguments arrive in blocks of eight scalars, and
  // a block that was spilled comes back whole for every use of one member)
  int nsprites = t.nsprites;
  asm volatile("" : "+s"(nsprites));
  const uint32_t magic_rows = kv.magic_rows;
  const uint32_t magic_p = kc.magic_p;
  const uint32_t magic_spw = kv.magic_spw;
  const int py = lane & 7, sub = lane >> 3;
  uint8_t* atlas_row = atlas + py * 32;
  const uint32_t scratch_off = (uint32_t)(lo.scratch - lo.atlas) + (uint32_t)(wave * t.scratch_cells) * 256u;

  // Copy phase geometry.  A pass's span (R strips x 8 pixel rows) is written as
  // 16-byte chunks, lane-contiguous: chunk q = bytes [16q, 16q + 16) of the span.
  // Rows are multiples of 8 bytes and cells are 3 x 8 bytes, so each half of a
  // chunk lies inside one cell's pixel row: key = cell << 8 | byte offset of the
  // half inside the cell's 256-byte packed image.  A half beyond the span names
  // cell 63: spans that are not whole KiBs have fewer than 64 cells (192 B each),
  // lane 63 is then a dead cell in every pass and its record says "no copy" — one
  // test per half instead of two.  The keys are the same in every pass.
  const uint32_t span_bytes = (uint32_t)R * 8u * row_bytes;
  const int n_iters = (int)((span_bytes + 1023u) >> 10);   // <= 12: at most 64 cells x 192 B
  const int n_full = (int)(span_bytes >> 10);              // chunks wholly inside the span (>= 6: 33 cells at least)
  uint32_t keys[12];
#pragma unroll
  for (int it = 0; it < 12; ++it) {
    uint32_t kk = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t pp = (uint32_t)(it * 64 + lane) * 16u + 8u * h;
      uint32_t key = 63u << 8;
      if (pp < span_bytes) {
        const uint32_t row = fast_div(pp, row_bytes, 1.0f / (float)row_bytes);
        const uint32_t colb = pp - row * row_bytes;
        const uint32_t ccx = fast_div(colb, 24u, 1.0f / 24.0f);
        key = (((row >> 3) * (uint32_t)row_cells + ccx) << 8) | ((row & 7u) * 32u + (colb - ccx * 24u));
      }
      kk |= key << (16 * h);
    }
    keys[it] = kk;
  }


  // One pass: strips [s0, s0 + R) of the batch whose records start at `wlds`.
  auto render_pass = [&](const uint32_t s0, const uint32_t nstrips, const uint8_t* wlds,
                         uint8_t* out_block) {
    // ---- phase 1 (lane = cell): resolve the draw list top -> bottom; a lane is
    // done at its first opaque sprite (everything below is hidden).  All plane
    // bytes are fetched first and all table entries second, so the pass pays two
    // LDS round trips instead of two per layer.
    int n_partial, n_ov;
    {
      const uint32_t strip = s0 + sr;
      const bool live = sr < R && strip < nstrips;
      const uint32_t sidx = live ? strip : 0u;
      const uint32_t img = magic_div(sidx, magic_rows);  // local world, or world*P + viewer
      const uint32_t cy = sidx - img * strip_rows;
      uint32_t lw = img, viewer = P, vo = 0;
      if (!wv) {
        lw = magic_div(img, magic_p);
        viewer = img - lw * P;
      }
      const uint8_t* grid = wlds + lw * wstride;
      const uint8_t* head = grid + t.grid_pad;  // ax[16] ay[16] aori[16] aalive[16]
      int cell;
      uint32_t oob_img = 0;   // (per-agent view: what a cell beyond the map shows this viewer)
      if (wv) {
        cell = (int)(cy * W + cx);
      } else {
        // (the viewer's four head bytes in ONE LDS round trip, the rotation as selects: the test of
        // `alive` used to stand between them)
        const uint32_t alive = head[48 + viewer], ori = head[32 + viewer];
        const int hx = head[viewer], hy = head[16 + viewer];
        oob_img = oobimg[viewer];
        const bool on_grid = alive != 0u;   // A6: an off-grid viewer sees only OutOfBounds
        vo = on_grid ? ori : 0u;
        const int dx = (int)cx - t.vl, dy = (int)cy - t.vf;  // right, down in view frame
        const int ax = vo == 0u ? dx : vo == 1u ? -dy : vo == 2u ? -dx : dy;
        const int ay = vo == 0u ? dy : vo == 1u ? dx : vo == 2u ? -dy : -dx;
        int x = hx + ax, y = hy + ay;
        bool inside;
        if (t.topology == 1) {
          // TORUS: the window reaches at most one map width / height beyond either
          // edge (mp_create checks it), so wrapping is one conditional add and one
          // conditional subtract — four integer modulos per lane and pass before
          x += x < 0 ? W : 0; x -= x >= W ? W : 0;
          y += y < 0 ? H : 0; y -= y >= H ? H : 0;
          inside = true;
        } else {
          inside = x >= 0 && x < W && y >= 0 && y < H;
        }
        cell = on_grid && inside ? y * W + x : -1;
      }
      const uint16_t* rinfo_v = rinfo + viewer * (uint32_t)nsprites;   // this viewer's sprite map
      CellRec r;
      r.base = 0; r.ov0 = 0; r.ov1 = 0; r.ov2 = 0;
      uint32_t base_img = 0;                         // image 0 is black
      bool done = !live || cell < 0;
      if (cell == -1) base_img = oob_img;   // (never in the world view: its dead lanes have !live)
      const uint16_t* tf = stab + (((0u - vo) & 3u) << 8);  // pieces other than avatars face north
      const uint8_t* gp = grid + (cell >= 0 ? cell : 0);
      // Round 6.  The resolve is written for the LDS round trips a pass pays.  Before: twelve
      // unrolled layers, `l < L` / avatar? / empty? / opaque? as nested tests — wave-uniform
      // branches between the layers' loads, so every plane byte and every table entry was a
      // dependent LDS round trip of its own (2 x 9 in a row for clean_up), `l < L` itself
      // carried as twelve lane masks spilled to VGPR lanes.  That chain, not the store path, set
      // the renderers' pace: the launch took 105 us on every buffer, 13 of them head, where its
      // own store loop takes 72 - 78 on a good one (profiles/r06_resolve.md).  Now: only the
      // planes that can show anything are read (FrameConsts::nvis / plane_off, from
      // DevTables::vis_layers — clean_up 7 of 9, commons_harvest 5 of 8, the matrix levels 4 of
      // 8: the logic layers' states have no sprite) by straight-line code unrolled for exactly
      // that count: all plane bytes in ONE round trip, all entries in a second; the opaque
      // search is selects on lane masks, no divergent region; the avatar look-ups run only in
      // a pass that holds an avatar.
      uint32_t base_e = base_img;
      auto resolve = [&](auto nv_tag) {
        constexpr int NV = decltype(nv_tag)::value;
        uint32_t offs[6] = {kv.plane_off[0], kv.plane_off[1], kv.plane_off[2],
                            kv.plane_off[3], kv.plane_off[4], kv.plane_off[5]};
        uint32_t avp = kv.av_planes;
        // (per pass: the bit fields are taken apart by scalar instructions where they are used,
        // not hoisted out of the ticket loop into twelve more spilled scalars)
#pragma unroll
        for (int i = 0; i < (NV + 1) / 2; ++i) asm volatile("" : "+s"(offs[i]));
        asm volatile("" : "+s"(avp));
        uint32_t ent[NV];
        uint32_t seen = 0;
#pragma unroll
        for (int k = 0; k < NV; ++k) ent[k] = gp[(offs[k >> 1] >> (16 * (k & 1))) & 0xffffu];
#pragma unroll
        for (int k = 0; k < NV; ++k) { ent[k] = tf[ent[k]]; seen |= ent[k]; }   // tf[0] == 0
        if (__ballot((seen & kAvatarBit) != 0u) != 0ull) {
          // avatars: own orientation, per-viewer sprite map
#pragma unroll
          for (int k = 0; k < NV; ++k) {
            if (!((avp >> k) & 1u)) continue;
            uint32_t e = ent[k];
            if (e & kAvatarBit) {
              const uint32_t si = sinfo[e & 255u];
              const uint32_t ori = head[32 + (si >> 8) - 1];
              const uint32_t rm = rinfo_v[si & 255u];
              e = ((rm >> 8) << 10) | slot[((rm & 255u) << 2) | ((ori - vo) & 3u)];
            }
            ent[k] = e;
          }
        }
#pragma unroll
        for (int k = NV - 1; k >= 0; --k) {          // top -> bottom
          const uint32_t e = ent[k];
          const bool opaque = (e & ((uint32_t)FLAG_OPAQUE << 10)) != 0u;
          base_e = (opaque && !done) ? e : base_e;
          done = done || opaque;
          if (e != 0u && !done) {                    // prepend: the list is kept bottom -> top
            r.ov2 = (r.ov2 << 12) | (r.ov1 >> 20);
            r.ov1 = (r.ov1 << 12) | (r.ov0 >> 20);
            r.ov0 = (r.ov0 << 12) | e;
          }
        }
      };
      switch (kv.nvis) {
        case 1: resolve(std::integral_constant<int, 1>()); break;
        case 2: resolve(std::integral_constant<int, 2>()); break;
        case 3: resolve(std::integral_constant<int, 3>()); break;
        case 4: resolve(std::integral_constant<int, 4>()); break;
        case 5: resolve(std::integral_constant<int, 5>()); break;
        case 6: resolve(std::integral_constant<int, 6>()); break;
        case 7: resolve(std::integral_constant<int, 7>()); break;
        case 8: resolve(std::integral_constant<int, 8>()); break;
        case 9: resolve(std::integral_constant<int, 9>()); break;
        case 10: resolve(std::integral_constant<int, 10>()); break;
        case 11: resolve(std::integral_constant<int, 11>()); break;
        case 12: resolve(std::integral_constant<int, 12>()); break;
        default: break;                              // (no plane shows anything)
      }
      base_img = base_e & 1023u;
      // composite cache: while the lowest overlay on the current base is a stack
      // the map's static pieces form (dirt on water, a shadow on sand ...), take
      // the pre-blended image as the base and drop the overlay
      if (t.pair_probe > 0) {
        for (int fold = 0; fold < 2 && r.ov0 != 0; ++fold) {
          const uint32_t key = (base_img << 10) | (r.ov0 & 1023u);
          uint32_t h = pair_hash(base_img, r.ov0 & 1023u), hit = 0;
          for (int k = 0; k < t.pair_probe; ++k) {
            const uint32_t ent = pairs[(h + k) & (kPairSlots - 1)];
            if ((ent >> 10) == key) { hit = ent & 1023u; break; }
            if (ent == 0xffffffffu) break;
          }
          if (hit == 0) break;
          base_img = hit;
          r.ov0 = (r.ov0 >> 12) | (r.ov1 << 20);
          r.ov1 = (r.ov1 >> 12) | (r.ov2 << 20);
          r.ov2 >>= 12;
        }
      }
      r.base = base_img * kSpriteStride;
      // 8-bit alpha somewhere in what is left of the list: FLAG_PARTIAL (bit 11) of its 12-bit entries
      const bool partial = ((r.ov0 & 0x00800800u) | (r.ov1 & 0x08008008u) | (r.ov2 & 0x80080080u)) != 0u;
      // cells with overlays go to a dense list, 8-bit-alpha ones first, so the
      // blend code below runs on full groups of lanes that all need it
      const bool has_ov = live && r.ov0 != 0;
      const unsigned long long mp = __ballot(has_ov && partial), mb = __ballot(has_ov && !partial);
      n_partial = __popcll(mp);
      n_ov = n_partial + __popcll(mb);
      if (has_ov) {
        const unsigned long long below = (1ull << lane) - 1ull;
        ovlist[partial ? __popcll(mp & below) : n_partial + __popcll(mb & below)] = (uint8_t)lane;
      }
      if (has_ov) r.base |= kSkipCopy;
      if (!live) r.base |= kSkipCopy | kDeadCell;
      recs[lane] = r;
    }

    FRAME_STAGE(20, n_ov);   // (developer timeline: phase 1 done)
    uint8_t* span = out_block + (size_t)s0 * 8 * row_bytes;
    {
      // the span base is wave-uniform: keep it in SGPRs (saddr form of the stores)
      const uint64_t sp = reinterpret_cast<uint64_t>(span);
      span = reinterpret_cast<uint8_t*>(
          ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32) |
          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sp));
    }

    // ---- phase 2a: every cell that shows a single opaque image — the bulk.
    // Opaque images are stored pre-packed (8 rows of 24 B RGB + 8 B pad), so the
    // span is assembled straight from the LDS atlas, two 8-byte reads per lane,
    // and leaves as full 16-byte lane-contiguous vectors (1 KiB per wave store:
    // whole cache lines, 2.6 x fewer L2 write requests than 12-byte row halves).
    // Halves that belong to a composited cell are left to phase 2b.
    // (Round 4, second session.  A wave issues one instruction every four cycles whatever its
    // kind, and a pass was ~570 vector + ~530 scalar + ~190 branch instructions (SQ counters,
    // profiles/r04_head.md): a store cost 18 - 20 instructions of bookkeeping — the sc1 / nt
    // choice, the `it >= n_iters` test through a spilled 64-bit mask, three EXEC-masked
    // regions for "both halves / the first / the second".  Now only the last two chunks are
    // tested against the span, and a chunk whose 128 halves are all plain single-image
    // cells — nearly every one — leaves behind ONE wave-uniform test.  (The store policy
    // chosen once per pass, two copies of this code: 13 - 23 VGPRs spilled; not kept.))
    // (Round 6: `plain` — wave-uniform: the pass is whole and every composited cell was staged, so
    // every cell a chunk inside the span touches shows ONE image in LDS.  Then the first six chunks
    // — 6 KiB: no span is shorter — leave with no test and no flag to mask, the others behind one
    // scalar compare against the span's whole KiBs.  A
    // pass was ~320 instructions of copy phase for twelve stores; this is ~130.)
    auto copy_cells = [&](auto plain_tag) {
      constexpr bool kPlain = decltype(plain_tag)::value;
      const bool kSc1 = sc1;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const bool kBare = kPlain && half == 0;
        uint32_t ba[6], bb[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const uint32_t kk = keys[half * 6 + i];
          ba[i] = recs[(kk >> 8) & 63u].base;
          bb[i] = recs[(kk >> 24) & 63u].base;
        }
        uint2 da[6], db[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const uint32_t kk = keys[half * 6 + i];
          da[i] = *reinterpret_cast<const uint2*>(atlas + (kBare ? ba[i] : (ba[i] & ~kSkipCopy)) + (kk & 255u));
          db[i] = *reinterpret_cast<const uint2*>(atlas + (kBare ? bb[i] : (bb[i] & ~kSkipCopy)) + ((kk >> 16) & 255u));
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int it = half * 6 + i;
          const uint32_t off = (uint32_t)(it * 64 + lane) * 16u;
          if (kBare || (kPlain && it < n_full)) {
            store_chunk<kNt>(span, off, da[i], db[i], kSc1);
            continue;
          }
          if (it >= 10 && it >= n_iters) break;   // (a span is 6.2 - 12 KiB: 33 - 64 cells x 192 B)
          if (__ballot(((ba[i] | bb[i]) & kSkipCopy) != 0u) == 0ull) {
            store_chunk<kNt>(span, off, da[i], db[i], kSc1);
            continue;
          }
          const bool oka = !(ba[i] & kSkipCopy);
          const bool okb = !(bb[i] & kSkipCopy);
          if (oka && okb) store_chunk<kNt>(span, off, da[i], db[i], kSc1);
          else if (oka) store_half<0, kNt>(span, off, da[i], kSc1);
          else if (okb) store_half<8, kNt>(span, off, db[i], kSc1);
        }
      }
    };

    // ---- phase 2b: the listed cells, eight per sub-pass (eight lanes per cell, one
    // per pixel row), composited in registers.  The first `scratch_cells` of them
    // are staged in LDS as one more pre-packed image each — the copy phase then
    // treats such a cell like any other; a pass with more composited cells than the
    // staging area holds stores the rest straight from the registers, two 12-byte
    // stores per row (their records keep kSkipCopy).
    for (int k0 = 0; k0 < n_ov; k0 += 8) {
      const int k = k0 + sub;
      if (k >= n_ov) continue;
      const int c = ovlist[k];
      const CellRec r = recs[c];
      const uint8_t* row = atlas_row + (r.base & ~kSkipCopy);
      const uint4 a = *reinterpret_cast<const uint4*>(row);
      const uint2 bb = *reinterpret_cast<const uint2*>(row + 16);
      uint32_t w[6] = {a.x, a.y, a.z, a.w, bb.x, bb.y};
      uint32_t acc[8];
      unpack_row(w, acc);
      uint32_t o0 = r.ov0, o1 = r.ov1, o2 = r.ov2;
      while (o0 != 0) {
        const uint32_t e = o0 & 4095u;
        o0 = (o0 >> 12) | (o1 << 20);
        o1 = (o1 >> 12) | (o2 << 20);
        o2 >>= 12;
        const uint8_t* orow = atlas_row + (e & 1023u) * kSpriteStride;
        if ((e >> 10) & FLAG_PARTIAL) blend_row<2>(acc, orow);
        else blend_row<1>(acc, orow);
      }
      pack_row(acc, w);
      const uint4 lo4 = {w[0], w[1], w[2], w[3]};
      const uint2 hi2 = {w[4], w[5]};
      if (k < t.scratch_cells) {
        const uint32_t img = scratch_off + (uint32_t)k * 256u;
        uint8_t* dst = atlas_row + img;
        *reinterpret_cast<uint4*>(dst) = lo4;
        *reinterpret_cast<uint2*>(dst + 16) = hi2;
        if (py == 0) recs[c].base = img;
      } else {
        store_row<kNt>(span, offtab[c] + (uint32_t)py * row_bytes, lo4, hi2, sc1);
      }
    }
    FRAME_STAGE(21, 0);      // (developer timeline: composited cells staged; the copy phase next)
    // (n_ov, the pass's extent and the store policy are wave-uniform)
#if defined(MP_NO_PLAIN_COPY)   // developer build: every pass takes the tested road (A/B of the bare one)
    const bool plain = false;
#else
    const bool plain = n_ov <= t.scratch_cells && s0 + (uint32_t)R <= nstrips;
#endif
    if (plain) copy_cells(std::true_type());
    else copy_cells(std::false_type());
  };

  // ---- the pipeline: tickets (batch, pass) of this wave's view, in order.  Lane 0
  // does the LDS bookkeeping of an iteration in ONE block — count the previous pass
  // done, take the next ticket — and the ticket is read back with v_readlane (lane 0,
  // whatever EXEC is).  Written as readfirstlane(lane == 0 ? atomicAdd() : 0)
  // next to a second `if (lane == 0)` further down the body, the compiler split
  // the loop body by "lane == 0 or not": lanes 1-63 then read ticket 0 forever.
  int prev_buf = -1;
  uint32_t* my_tickets = &ctrl->next_ticket[wv ? 1 : 0];
  for (;;) {
    // the previous pass's LDS reads have returned (its stores may still be in flight)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint32_t taken = 0;
    if (lane == 0) {
      if (prev_buf >= 0) atomicAdd(&ctrl->done[prev_buf], 1u);
      taken = atomicAdd(my_tickets, 1u);
    }
    const uint32_t ticket = (uint32_t)__builtin_amdgcn_readlane((int)taken, 0);
    FRAME_STAGE(7, ticket);
    // (ticket / npb and k % NB by the host's reciprocals: a launch hands out thousands of tickets
    // per workgroup at most — exact while ticket * npb < 2^32, frame_consts)
    const int k = (int)magic_div(ticket, kv.magic_npb);
    prev_buf = -1;
    bool stalled = false;
    const int w0 = batch_first_world(k, stalled);
    if (stalled) break;
    if (w0 < 0) {   // no such batch: the launch is over once every chain's pool has run dry
      if (all_chains_ended(k)) break;
      continue;
    }
    const uint32_t s0 = (ticket - (uint32_t)k * npb) * (uint32_t)R;
    int nw = N - w0;
    if (nw > B) nw = B;
    const uint32_t nstrips = (uint32_t)(nw * strips_per_world);
    const int kb_now = k - (int)magic_div((uint32_t)k, K.magic_nb) * NB;   // k % NB
    const int r0 = kb_now * B;
    {
      // the worlds this pass reads (strips [s0, s0 + R) of batch k) are in ring buffer
      // k % NB: slots [first, last] — a WORLD.RGB pass touches one or two worlds, so
      // drawing starts when the FIRST world of a batch is published, not the last
      const uint32_t want = (uint32_t)(k + 1);
      uint32_t last_strip = s0 + (uint32_t)R - 1u;
      if (last_strip >= nstrips) last_strip = nstrips - 1u;
      const uint32_t first = magic_div(s0 < nstrips ? s0 : 0u, magic_spw);
      const uint32_t last = magic_div(last_strip, magic_spw);
      uint64_t wait_t0 = 0;
      for (uint32_t polls = 0;; ++polls) {
        const uint32_t v = ((uint32_t)lane >= first && (uint32_t)lane <= last)
                               ? lds_acquire(&ctrl->slot_batch[r0 + lane]) : want;
        const unsigned long long late = __ballot(v != want);
        if (late == 0) break;
        if (waited_too_long(polls, wait_t0)) {
          report_stall(t, lane, FAULT_BATCH_READY, (uint32_t)wave, (uint32_t)k,
                       (uint32_t)late, want);
          stalled = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (stalled) break;
    }
    FRAME_STAGE(8, ticket);
#if defined(MP_FRAME_ENDS)
    // developer build: when did this workgroup draw its first pass (tools/gpu_frame_ends.py)
    if (lane == 0 && ticket == 0) t.claim[2 + 2 * blockIdx.x] = (uint32_t)wall_clock64();
#endif
    if (s0 < nstrips)
      render_pass(s0, nstrips, smem + lo.records + r0 * wstride,
                  out + (size_t)w0 * strips_per_world * 8 * row_bytes);
    prev_buf = kb_now;
    for (int i = 0; i < pace; ++i) __builtin_amdgcn_s_sleep(8);
    FRAME_STAGE(9, ticket);
  }
  FRAME_STAGE(14, 0);
#if defined(MP_FRAME_ENDS)
  // ... and when did its last renderer wave run out of tickets (the max over the waves)
  if (lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    atomicMax(&t.claim[2 + 2 * blockIdx.x + 1], (uint32_t)wall_clock64());
  }
#endif
}

}  // namespace

// Launch geometry.  One workgroup per CU (all 160 KB of LDS): the sprite
// atlas and tables are staged once per CU, every wave has its staging area for
// composited cells, and a ring of NB buffers of B worlds each takes the rest: enough
// slots that drawing the resident worlds (tens of us) covers the feeders' steps of
// the next ones (~10 us each, in parallel), few enough that they fit.
static int slot_scratch_bytes(const DevTables& t, const SubstrateTables& s) {
  int extra = 0;
  if (s.substrate == MPK_SUBSTRATE_TERRITORY) extra = stepk::extra_bytes(s.tr);
  if (s.substrate == MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS) extra = stepk::extra_bytes(s.em);
  return stepk::scratch_bytes(t) + extra;
}

static int gcd_int(int a, int b) { while (b) { const int r = a % b; a = b; b = r; } return a; }

// views: 0 = per-agent RGB, 1 = WORLD.RGB, 2 = both in one launch
FramePlan plan_frame(const DevTables& t, const SubstrateTables& s, int num_worlds,
                     bool with_step, int views, int num_cus, const MpDevOptions* dev) {
  FramePlan p = {};
  const bool world_view = views == 1;
  const int max_waves = ((with_step && s.substrate == MPK_SUBSTRATE_THE_MATRIX) ? kMatrixThreads
                                                                                : kDrawThreads) / 64;
  // Renderers: the drawing is the store path's business, and more waves are not
  // better — WORLD.RGB (720-byte rows) is drawn fastest by 8 waves (96 us; 110 us
  // with 12, 113 us with 10, same box), the per-agent views (264-byte rows) by
  // 12-13.  Feeders: a step takes 10-25 us of one wave (a chain of dependent LDS
  // and scalar round trips) and a CU's 16-32 worlds must be fed faster than they
  // are drawn: measured, fused clean_up 259 / 176 / 134 us with 1 / 2 / 4 feeders
  // (profiles/r02_frame_geometry.md, profiles/r02_frame_timeline.md)
  p.nwaves = world_view ? 12 : 16;
  p.feeders = 4;
  int B = 4, NB = 2;
  // per-agent views, fused: batches of three leave the composite cache more LDS
  // and draw faster whatever the box.  Feeders: since round 3 a feeder's later
  // steps run at the renderers' priority and its record reads hit the cache (nt
  // pixel stores); fewer feeders = more drawing waves.  commons_harvest: 313 us
  // with 6 feeders on every box; with 3 (13 drawing waves, four per SIMD) 266 us
  // on one box and 350 on three others — the per-agent drawing is then issue-bound
  // and the fourth wave of a SIMD starves (pass times 4.8 / 5.3 / 5.9 / 7.2 us by
  // wave on a fast CU, 5.6 / 6.5 / 9.5 / 12.6 on a slow one of the same launch):
  // 6 stays.  territory, whose step is 3 x longer: 366 us with 3, 434 with 2 or 6;
  // the matrix level 6 feeders at priority 3 (329 us; 362-397 with 3)
  // (tools/gpu_r03_call10.sh / call13.sh, profiles/r03_frame_plans.md)
  if (with_step && !world_view && max_waves == 16) {
    B = 3;
    p.feeders = s.substrate == MPK_SUBSTRATE_THE_MATRIX ? 6
                : s.substrate == MPK_SUBSTRATE_TERRITORY ? 3
                : 6;
    // small views (the two-player games: under 64 KB of pixels a world): a CU has
    // 64 worlds to step for a few us of drawing each, the stepping is the long
    // pole: batches of 8 and 8 feeders (matrix games: 102 us against 114 in two
    // launches, 125-139 with batches of 4-6, 152 with 4 feeders), 4 feeders for
    // coins' short step (156 us against 190 in two launches, 176 with 8)
    const long long view_bytes = (long long)t.P * (t.vf + t.vb + 1) * (t.vl + t.vr + 1) *
                                 t.sprite_size * t.sprite_size * 3;
    // (round 5: the 40 x 40 views of two players — collaborative_cooking's small kitchens,
    // 9.6 KB a world — leave the renderers next to nothing to do: 8 feeders 26.0 us, 4 feeders
    // 36.7; the nine-player kitchen, 43 KB a world, is indifferent, 60.7 / 61.1:
    // tools/history/gpu_r05_call17.sh)
    if (view_bytes < 64 * 1024 && views == 0) {
      B = 8;
      p.feeders = (s.substrate == MPK_SUBSTRATE_THE_MATRIX || view_bytes < 16 * 1024) ? 8 : 4;
    } else if (view_bytes < 150 * 1024 && views == 0 && s.substrate != MPK_SUBSTRATE_THE_MATRIX &&
               s.substrate != MPK_SUBSTRATE_TERRITORY) {
      // (round 5: five or six viewers of 88 x 88 — coop_mining, gift_refinements,
      // externality_mushrooms.  Batches of 4 with 4 feeders: 4096 worlds are then 256
      // workgroups x 4 batches, where batches of 3 are 228 x 6 and leave 28 CUs idle; same
      // buffers, tools/history/gpu_r05_call28.sh: 111.6 against 130.2 - 132.0 us, 112.8 against
      // 121.4 - 126.1, 104.7 against 114.6 - 115.8; clean_up's seven viewers: 146.6 against 145.1)
      B = 4;
      p.feeders = 4;
    }
  }
  // The same for what `substrate.build` binds on the small substrates (round 5, same buffers,
  // tools/history/gpu_r05_call20.sh): BOTH views with per-agent views under 45 KB a world —
  // batches of 8 and 8 feeders: collaborative_cooking cramped 45.5 -> 34.7 us, crowded 97.7 ->
  // 87.8, prisoners_dilemma repeated 106.4 -> 88.1 (coins, 46.5 KB: 93.2 -> 100.3, stays); and
  // WORLD.RGB alone under 16 KB a world (the two-player kitchens): 34.8 -> 26.2 (the matrix
  // games' and coins' world views are 66 KB and more: 74.9 -> 123, 53.3 -> 90 — they stay)
  if (with_step && max_waves == 16) {
    const long long agent_bytes = (long long)t.P * (t.vf + t.vb + 1) * (t.vl + t.vr + 1) *
                                  t.sprite_size * t.sprite_size * 3;
    const long long world_bytes = (long long)t.H * t.W * t.sprite_size * t.sprite_size * 3;
    if ((views == 2 && agent_bytes < 45 * 1024) || (views == 1 && world_bytes < 16 * 1024)) {
      B = 8;
      p.feeders = 8;
    } else if (views == 1 && world_bytes < 64 * 1024) {
      // WORLD.RGB alone, 16 - 64 KB a world (the two larger kitchens, externality_mushrooms):
      // still the stepping that takes the time — 16 waves, half of them feeders (same
      // buffers, tools/history/gpu_r05_call33.sh: crowded 54.6 -> 39.9 us, figure_eight
      // 48.2 -> 37.4 with batches of 8; externality_mushrooms, 62 KB, 73.9 -> 67.8 with 6)
      B = world_bytes < 32 * 1024 ? 8 : 6;
      p.feeders = B;
      p.nwaves = 16;
    }
  }
  if (p.nwaves > max_waves) p.nwaves = max_waves;
  p.slot_scratch = with_step ? slot_scratch_bytes(t, s) : 0;
  if (num_cus <= 0) num_cus = 1;
  // feeders after their first world: back to the renderers' priority, except for
  // the long steps (territory 433 us at priority 0, 371 us at 3; clean_up 119 -> 115,
  // commons 327 -> 309 the other way round)
  p.late_prio = (s.substrate == MPK_SUBSTRATE_TERRITORY || s.substrate == MPK_SUBSTRATE_THE_MATRIX) ? 3 : 0;
  int static_pct = 100;
  // test / development overrides (MpConfig.dev: test_frame_geometry_edge_cases,
  // tools/gpu_plan_sweep.sh); NULL in product paths
  if (dev) {
    if (dev->late_feeder_prio > 0) p.late_prio = dev->late_feeder_prio - 1;
    if (dev->batch_worlds > 0) B = dev->batch_worlds;
    if (dev->ring_batches > 0) NB = dev->ring_batches;
    if (dev->waves > 0) p.nwaves = dev->waves;
    if (dev->feeders > 0) p.feeders = dev->feeders;
    if (dev->max_groups > 0 && dev->max_groups < num_cus) num_cus = dev->max_groups;
    if (dev->static_pct > 0) static_pct = dev->static_pct > 100 ? 100 : dev->static_pct;
  }
  if (p.nwaves < (views == 2 ? 3 : 2)) p.nwaves = views == 2 ? 3 : 2;   // a feeder + a renderer per view
  if (p.nwaves > max_waves) p.nwaves = max_waves;
  if (B > kMaxBatch) B = kMaxBatch;
  if (B > num_worlds) B = num_worlds;
  if (NB < 2) NB = 2;
  while (NB * B > kMaxSlots && NB > 2) --NB;
  // F divides the ring (NB * B slots), leaves a wave to draw, and its claim chains
  // (A = F / gcd(F, B)) fit the buffers: a buffer serves ONE chain (A divides NB)
  auto fit_feeders = [&]() {
    if (p.feeders > NB * B) p.feeders = NB * B;
    if (p.feeders > p.nwaves - (views == 2 ? 2 : 1)) p.feeders = p.nwaves - (views == 2 ? 2 : 1);
    for (; p.feeders > 1; --p.feeders) {
      const int chains = p.feeders / gcd_int(p.feeders, B);
      if ((NB * B) % p.feeders == 0 && NB % chains == 0 && chains <= kMaxChains) break;
    }
  };
  fit_feeders();
  while (frame_lds_layout(t, NB * B, p.feeders, p.nwaves, p.slot_scratch).total > 160 * 1024) {
    if (NB > 2) --NB;
    else if (B > 1) --B;
    else break;
    fit_feeders();
  }
  // (a developer override of the staging area can still be too big: fewer waves)
  while (p.nwaves > 4 &&
         frame_lds_layout(t, NB * B, p.feeders, p.nwaves, p.slot_scratch).total > 160 * 1024) {
    --p.nwaves;
    fit_feeders();
  }
  p.B = B;
  p.NB = NB;
  p.store_sc1 = (dev && dev->store_sc1 > 0) ? 1 : 0;
  p.pace = (dev && dev->pace > 0) ? dev->pace - 1 : 0;
  p.team = 0;   // (set below, once the split is known)
  p.head = with_step ? kStockHead : 0;
  if (with_step && dev && dev->head > 0) p.head = (dev->head - 1) & 1;
  // two views: the renderer waves are shared out by the bytes each view writes
  p.world_waves = 0;
  if (views == 2) {
    const int renderers = p.nwaves - p.feeders;
    const long long wb = (long long)t.H * t.W, ab = (long long)t.P * (t.vf + t.vb + 1) * (t.vl + t.vr + 1);
    int ww = (int)((renderers * wb + (wb + ab) / 2) / (wb + ab));
    if (dev && dev->world_waves > 0) ww = dev->world_waves;
    if (ww < 1) ww = 1;
    if (ww > renderers - 1) ww = renderers - 1;
    p.world_waves = ww;   // (fit_feeders leaves two renderers: one per view at least)
  }
  // the worlds: an even split of whole batches over the workgroups (whole batches,
  // except in the last workgroup: territory 249 workgroups x 33 worlds rather than
  // 256 x 32 with a partial eleventh batch each: measured, 408 vs 414 us; filling all
  // 256 CUs instead — commons_harvest 256 x 16 with a ragged sixth batch rather than
  // 228 x 18 — is 6 % SLOWER over eight buffers, profiles/r03_buffer_placement.md) —
  // or (static_pct < 100) a smaller even split and the rest in the pool
  const int nbt = (num_worlds + B - 1) / B;
  int groups = nbt < num_cus ? nbt : num_cus;
  const int fair = (nbt + groups - 1) / groups;
  const int chains = p.feeders / gcd_int(p.feeders, B);
  int ks = fair;
  if (static_pct < 100) {
    ks = fair * static_pct / 100;
    if (ks < chains) ks = chains;   // a chain's first claim rides on an owned batch
  }
  if (ks >= fair) {
    p.ks = fair;
    p.groups = (nbt + fair - 1) / fair;
    p.pool = 0;
  } else {
    p.ks = ks;
    p.groups = groups;              // every CU: nbt >= groups * fair - (groups - 1) > groups * ks
    if ((long long)p.groups * ks > nbt) p.groups = nbt / ks;
    p.pool = nbt - p.groups * ks;
  }
  // XCD teams (MpDevOptions.team; mp_tune times it as a candidate): single-world batches, nothing
  // pooled (a pass then never spans two worlds of a batch that do not lie next to each other)
  if (dev && dev->team > 0 && p.B == 1 && p.pool == 0) p.team = 1;
  return p;
}

int frame_lds_bytes(const DevTables& t, const FramePlan& p) {
  return frame_lds_layout(t, p.NB * p.B, p.feeders, p.nwaves, p.slot_scratch).total;
}

// The world-independent part of a workgroup's LDS image (bytes [0, world) of
// frame_lds_layout), built once on the host.  `sprite_flags8`, `state_sprite`,
// `state_player`, `view_sprite_map`, `state_orient`, `img_slot`, `images` and
// `pair_table` are host copies of the tables of the same names.
int render_blob_bytes(const DevTables& t) { return frame_lds_layout(t, 1, 1, 1, 0).world; }

void build_render_blob(const DevTables& t, const uint8_t* images, const uint16_t* img_slot,
                       const uint32_t* pair_table, const int32_t* state_sprite,
                       const int8_t* state_player, const int32_t* view_sprite_map,
                       const uint8_t* sprite_flags8, const int32_t* state_orient,
                       uint8_t* blob) {
  const FrameLds lo = frame_lds_layout(t, 1, 1, 1, 0);
  memset(blob, 0, (size_t)lo.world);
  for (int i = 0; i < t.n_images; ++i)
    memcpy(blob + lo.atlas + (size_t)i * kSpriteStride, images + (size_t)i * 256, 256);
  uint16_t* sinfo = reinterpret_cast<uint16_t*>(blob + lo.sinfo);   // sprite | (player+1) << 8
  uint16_t* rinfo = reinterpret_cast<uint16_t*>(blob + lo.rinfo);   // remapped sprite | flags << 8
  uint16_t* slot = reinterpret_cast<uint16_t*>(blob + lo.slot);     // atlas image of (sprite, facing)
  uint16_t* stab = reinterpret_cast<uint16_t*>(blob + lo.stab);     // entry of (facing, state)
  uint32_t* pairs = reinterpret_cast<uint32_t*>(blob + lo.pairs);
  for (int s = 0; s < 256; ++s) {
    const int sp = s < t.nstates ? state_sprite[s] : -1;
    const int pl = s < t.nstates ? state_player[s] : -1;
    sinfo[s] = (uint16_t)((sp < 0 ? 0xff : sp) | ((pl + 1) << 8));
  }
  for (int i = 0; i < (t.P + 1) * t.nsprites; ++i) {
    const int sp = view_sprite_map[i];
    rinfo[i] = (uint16_t)(sp | ((sprite_flags8[sp] & 3) << 8));
  }
  for (int i = 0; i < t.nsprites * 4; ++i) slot[i] = img_slot[i];
  // state -> entry under the world sprite map, per relative facing; avatar
  // states are resolved per viewer (own orientation, Self remap) in phase 1
  for (int i = 0; i < 4 * 256; ++i) {
    const int f = i >> 8, st = i & 255;
    uint32_t e = 0;
    if (st < t.nstates && state_sprite[st] >= 0) {
      if (state_player[st] >= 0) {
        e = kAvatarBit | (uint32_t)st;
      } else {
        const int sp = view_sprite_map[t.P * t.nsprites + state_sprite[st]];
        // (a beam pseudo-state of an oriented sprite carries its own facing)
        e = ((uint32_t)(sprite_flags8[sp] & 3) << 10) | img_slot[sp * 4 + ((f + state_orient[st]) & 3)];
        // a sprite without a visible pixel (territory's level-1 marking) draws nothing
        if (sprite_flags8[sp] & 4) e = 0;
      }
    }
    stab[i] = (uint16_t)e;
  }
  for (int i = 0; i < kPairSlots; ++i) pairs[i] = pair_table[i];
  // what a cell beyond the map shows to viewer v: its OutOfBounds sprite (sprite 0 under its
  // sprite map), facing north — one look-up in phase 1 instead of two dependent ones
  uint16_t* oobimg = reinterpret_cast<uint16_t*>(blob + lo.oobimg);
  for (int v = 0; v <= t.P; ++v) oobimg[v] = slot[(rinfo[v * t.nsprites] & 255u) << 2];
}


// DevTables::vis_layers from the blob's state table: bit l = some state of render plane l has
// an entry (a sprite with a visible pixel, or an avatar's), bit 16 + l = some state of it is an
// avatar's.  `state_layer` is the host copy of the table of that name.
uint32_t render_visible_layers(const DevTables& t, const uint8_t* blob, const int32_t* state_layer) {
  const FrameLds lo = frame_lds_layout(t, 1, 1, 1, 0);
  const uint16_t* stab = reinterpret_cast<const uint16_t*>(blob + lo.stab);
  uint32_t vis = 0;
  for (int st = 0; st < t.nstates && st < 256; ++st) {
    const int l = state_layer[st];
    if (l < 0 || l >= t.L || l >= kMaxLayers) continue;
    for (int f = 0; f < 4; ++f) {
      const uint32_t e = stab[f * 256 + st];
      if (e != 0) vis |= 1u << l;
      if (e & kAvatarBit) vis |= 1u << (16 + l);
    }
  }
  return vis;
}

namespace {

// Everything a launch derives from its plan (FrameConsts): the divisions, on the host.
FrameConsts frame_consts(const DevTables& t, const FramePlan& p, int num_worlds, bool with_step) {
  FrameConsts K = {};
  K.p = p;
  K.lo = frame_lds_layout(t, p.NB * p.B, p.feeders, p.nwaves, p.slot_scratch);
  K.N = num_worlds;
  K.nbt = (num_worlds + p.B - 1) / p.B;
  K.chains = p.feeders / gcd_int(p.feeders, p.B);
  K.pool_first = p.groups * p.ks;
  K.b_mod_f = p.B % p.feeders;
  K.tables_vec = with_step ? stepk::tables_bytes(t) >> 4 : 0;
  K.record_vec = t.world_stride >> 4;
  for (int f = 0; f < p.feeders && f < 16; ++f)
    K.first_k_nibbles[f >> 3] |= (uint32_t)((f / p.B) & 15) << (4 * (f & 7));
  const int VW = t.vl + t.vr + 1, VH = t.vf + t.vb + 1;
  const int rc[2] = {VW, t.W}, sr[2] = {VH, t.H}, spw[2] = {t.P * VH, t.H};
  for (int v = 0; v < 2; ++v) {
    K.row_cells[v] = rc[v];
    K.strip_rows[v] = sr[v];
    K.R[v] = 64 / rc[v];
    K.strips_per_world[v] = spw[v];
    K.npb[v] = (uint32_t)((p.B * spw[v] + K.R[v] - 1) / K.R[v]);
    K.magic_rows[v] = div_magic((uint32_t)sr[v]);
    K.magic_spw[v] = div_magic((uint32_t)spw[v]);
    K.magic_npb[v] = div_magic(K.npb[v]);
  }
  K.magic_nb = div_magic((uint32_t)p.NB);
  K.magic_p = div_magic((uint32_t)t.P);
  for (int l = 0; l < kMaxLayers && l < t.L; ++l) {
    if (!((t.vis_layers >> l) & 1u)) continue;
    K.plane_off[K.nvis >> 1] |= (uint32_t)(l * t.H * t.W) << (16 * (K.nvis & 1));   // (< 65536: mp_create)
    if ((t.vis_layers >> (16 + l)) & 1u) K.av_planes |= 1u << K.nvis;
    ++K.nvis;
  }
  return K;
}

template <class Tables, class Sites>
void launch_one(const DevTables& t, const Tables& c, const stepk::StepArgs& args, uint8_t* out_a,
                uint8_t* out_w, const FramePlan& p, hipStream_t stream) {
  FrameConsts K = frame_consts(t, p, args.num_worlds, !std::is_same<Tables, NoTables>::value);
  const size_t lds = (size_t)K.lo.total;
  if (out_a && out_w) {
    K.npb_all = K.npb[0] + K.npb[1];
    hipLaunchKernelGGL((k_frame<Tables, Sites, 2>), dim3(p.groups), dim3(p.nwaves * 64), lds,
                       stream, t, c, args, out_a, out_w, K);
  } else if (out_w) {
    K.npb_all = K.npb[1];
    hipLaunchKernelGGL((k_frame<Tables, Sites, 1>), dim3(p.groups), dim3(p.nwaves * 64), lds,
                       stream, t, c, args, out_a, out_w, K);
  } else {
    K.npb_all = K.npb[0];
    hipLaunchKernelGGL((k_frame<Tables, Sites, 0>), dim3(p.groups), dim3(p.nwaves * 64), lds,
                       stream, t, c, args, out_a, out_w, K);
  }
}

template <class Tables, class Sites>
int allow_lds() {
  hipError_t r[3] = {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame<Tables, Sites, 0>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame<Tables, Sites, 1>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame<Tables, Sites, 2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)};
  for (hipError_t e : r)
    if (e != hipSuccess) return (int)e;
  return 0;
}

}  // namespace

// The frame kernels use up to 160 KB of dynamic LDS; declare it (a no-op where
// the runtime grants it anyway).  Called once per engine, with its device current.
int prepare_frame() {
  int rc = allow_lds<NoTables, NoSites>();
  if (!rc) rc = allow_lds<CleanUpTables, stepk::CleanUpSites>();
#if !defined(MP_FRAME_ISA_SUBSET)
  if (!rc) rc = allow_lds<CommonsTables, stepk::CommonsSites>();
  if (!rc) rc = allow_lds<TerritoryTables, stepk::TerritorySites>();
  if (!rc) rc = allow_lds<CoinsTables, stepk::CoinsSites>();
  if (!rc) rc = allow_lds<MatrixTables, stepk::MatrixSites>();
  if (!rc) rc = allow_lds<CoopTables, stepk::CoopSites>();
  if (!rc) rc = allow_lds<GiftTables, stepk::GiftSites>();
  if (!rc) rc = allow_lds<CookTables, stepk::CookSites>();
  if (!rc) rc = allow_lds<MushroomTables, stepk::MushroomSites>();
#endif
  return rc;
}

// One launch: the views `out_a` (per-agent RGB) and / or `out_w` (WORLD.RGB) of all
// worlds — from the records in HBM (s == NULL: mp_observe, views of a reset that
// names no world ...), or stepped first (one environment step or reset of all worlds
// + the views of the result).  `p` is the plan for exactly these views; p.parity
// alternates between consecutive frame launches of an engine (DevTables::claim).
void launch_frame(const DevTables& t, const SubstrateTables* s, const stepk::StepArgs& args,
                  uint8_t* out_a, uint8_t* out_w, const FramePlan& p, hipStream_t stream) {
  if (!s) {
    launch_one<NoTables, NoSites>(t, NoTables(), args, out_a, out_w, p, stream);
    return;
  }
#if defined(MP_FRAME_ISA_SUBSET)
  // developer build (tools/isa_stats.sh quick): the draw-only and the clean_up kernels alone
  if (s->substrate == MPK_SUBSTRATE_CLEAN_UP)
    launch_one<CleanUpTables, stepk::CleanUpSites>(t, s->cu, args, out_a, out_w, p, stream);
#else
  switch (s->substrate) {
    case MPK_SUBSTRATE_CLEAN_UP:
      launch_one<CleanUpTables, stepk::CleanUpSites>(t, s->cu, args, out_a, out_w, p, stream);
      break;
    case MPK_SUBSTRATE_COMMONS_HARVEST:
      launch_one<CommonsTables, stepk::CommonsSites>(t, s->ch, args, out_a, out_w, p, stream);
      break;
    case MPK_SUBSTRATE_TERRITORY:
      launch_one<TerritoryTables, stepk::TerritorySites>(t, s->tr, args, out_a, out_w, p, stream);
      break;
    case MPK_SUBSTRATE_COINS:
      launch_one<CoinsTables, stepk::CoinsSites>(t, s->co, args, out_a, out_w, p, stream);
      break;
    case MPK_SUBSTRATE_THE_MATRIX:
      launch_one<MatrixTables, stepk::MatrixSites>(t, s->mx, args, out_a, out_w, p, stream);
      break;
    case MPK_SUBSTRATE_COOP_MINING:
      launch_one<CoopTables, stepk::CoopSites>(t, s->cm, args, out_a, out_w, p, stream);
      break;
    case MPK_SUBSTRATE_GIFT_REFINEMENTS:
      launch_one<GiftTables, stepk::GiftSites>(t, s->gr, args, out_a, out_w, p, stream);
      break;
    case MPK_SUBSTRATE_COLLABORATIVE_COOKING:
      launch_one<CookTables, stepk::CookSites>(t, s->cc, args, out_a, out_w, p, stream);
      break;
    case MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS:
      launch_one<MushroomTables, stepk::MushroomSites>(t, s->em, args, out_a, out_w, p, stream);
      break;
  }
#endif
}
