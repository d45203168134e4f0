// render.hip — layer views + tile renderer for N worlds.
//
// Replaces the observation reads that follow every reference step
// (api:observation, lua/modules/api_factory.lua:73-75):
//   "N.RGB"      playerLayerView:observation -> playerView:render
//                (avatar_library.lua:225-277)   egocentric 11x11 cells, 88x88x3
//   "WORLD.RGB"  worldView:render(worldLayerView:observation)
//                (base_simulation.lua:347-368)  whole map, H*8 x W*8 x 3
// i.e. dmlab2d's `world:createView` + `tile.Scene:render`.  Semantics (window,
// rotation, OutOfBounds, per-viewer spriteMap, relative facing, bottom->top
// 8-bit alpha compositing) are the ones the CPU restatement in oracle/render.c
// documents as assumptions A6-A9; this file is bit-exact with it.
//
// Execution shape (v5; the measurements that led here are in
// profiles/r01_render_ablation.md).  The kernel writes 192 B per output cell and
// reads ~9 B, so it is HBM-write bound by construction; everything is arranged
// so that nothing but the stores touches the vector-memory pipe in steady
// state:
//   * a workgroup owns a few whole worlds.  Its prologue stages everything it
//     will ever read — the de-duplicated sprite atlas, the lookup tables and the
//     grid planes + avatar header of its worlds — into LDS.  After that the
//     main loop issues NO global loads: on gfx9-family parts loads and stores
//     share vmcnt and the per-CU memory pipe is in-order, so a load issued
//     behind a wave's stores waits for them to drain (measured: 165 us instead
//     of 90 us for the same stores);
//   * work unit = a "strip": one row of output cells = 8 pixel rows, which is
//     contiguous in the output tensor in both views.  A wave owns
//     floor(64 / row_cells) whole strips per pass, so a pass writes one
//     contiguous 64-byte-aligned span and completes every cache line itself;
//   * phase 1, one lane per cell: resolve the cell's draw list from the LDS
//     planes — bottom -> top, restarted at every fully opaque sprite so hidden
//     layers cost nothing — into a 16-byte record;
//   * phase 2, eight lanes per cell (one per pixel row): each lane composites
//     one 8-pixel row from the LDS atlas and stores its 24 bytes.  Control flow
//     diverges only between the 8 cells of a sub-pass, so the 8-bit alpha blend
//     (the expensive path) is paid only where such a sprite is on screen.
#include <stdlib.h>

#include "mp_common.h"

namespace {

constexpr int kMaxLayers = 12;
constexpr int kSpriteStride = 272;  // 8*8*4 B + 16 B pad: spreads images over LDS banks
constexpr int kHeadBytes = 64;      // WorldTail head: ax[16], ay[16], aori[16], aalive[16]
constexpr int kThreads = 512;       // 8 waves share one staged atlas: 32 waves/CU at 4 workgroups/CU
constexpr int kWaves = kThreads / 64;

enum { FLAG_OPAQUE = 1, FLAG_PARTIAL = 2 };

struct RenderLds { int atlas, sinfo, rinfo, slot, world, recs, offtab, total; };

__host__ __device__ inline RenderLds render_lds_layout(const DevTables& t, int wpb) {
  RenderLds r;
  int off = 0;
  r.atlas = off; off += t.n_images * kSpriteStride;
  r.sinfo = off; off += 256 * 2;                                    // u16 per state
  r.rinfo = off; off += (((t.P + 1) * t.nsprites * 2) + 15) & ~15;  // u16 per (viewer, sprite)
  r.slot = off; off += ((t.nsprites * 4 * 2) + 15) & ~15;           // u16 per (sprite, facing)
  r.world = off; off += wpb * (t.grid_pad + kHeadBytes);
  r.recs = off; off += kWaves * 64 * 16;                                 // per-wave draw lists
  r.offtab = off; off += 64 * 4;
  r.total = off;
  return r;
}

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

// 8 RGB pixels -> 24 bytes (dst is 8-byte aligned).
__device__ inline void store_row(uint8_t* dst, const uint32_t* px) {
  const uint32_t w0 = px[0] | (px[1] << 24);
  const uint32_t w1 = (px[1] >> 8) | (px[2] << 16);
  const uint32_t w2 = (px[2] >> 16) | (px[3] << 8);
  const uint32_t w3 = px[4] | (px[5] << 24);
  const uint32_t w4 = (px[5] >> 8) | (px[6] << 16);
  const uint32_t w5 = (px[6] >> 16) | (px[7] << 8);
  // Two 12-byte stores (the form hipcc picks for a plain 24-byte struct copy
  // in tools/ubench/store_bw2.hip, which reaches 5.5 TB/s).  Nothing ever
  // waits on these stores, so no vmcnt bookkeeping is needed around the asm.
  typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
  const u32x3 lo = {w0, w1, w2}, hi = {w3, w4, w5};
  asm volatile("global_store_dwordx3 %0, %1, off\n\t"
               "global_store_dwordx3 %0, %2, off offset:12"
               :: "v"(dst), "v"(lo), "v"(hi) : "memory");
}

// Composite one sprite row (8 px) onto the row held in registers.
template <int kMode>  // 0: opaque copy (alpha pre-cleared), 1: binary alpha, 2: 8-bit blend
__device__ inline void blend_row(uint32_t* acc, const uint8_t* row) {
  const uint4* src = reinterpret_cast<const uint4*>(row);
  const uint4 a = src[0], b = src[1];
  const uint32_t s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (kMode == 0) acc[j] = s[j];
    else if (kMode == 1) acc[j] = (s[j] >> 24) ? (s[j] & 0xffffffu) : acc[j];
    else acc[j] = blend_partial(acc[j], s[j]);
  }
}

__device__ inline uint32_t fast_div(uint32_t n, uint32_t d, float rcp) {
  uint32_t q = (uint32_t)((float)n * rcp);
  if (q * d > n) --q;
  else if ((q + 1) * d <= n) ++q;
  return q;
}

struct CellRec { uint64_t la, lb; };  // draw list of one output cell
constexpr uint64_t kNoCell = ~0ull;

template <bool kWorldView>
__global__ __launch_bounds__(kThreads) void k_render(DevTables t,
                                                const uint8_t* __restrict__ state,
                                                uint8_t* __restrict__ out,
                                                int num_worlds, int wpb, int ablate) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const RenderLds lo = render_lds_layout(t, wpb);
  const int tid = threadIdx.x;
  const int HW = t.H * t.W, L = t.L, P = t.P, W = t.W, H = t.H;
  uint8_t* atlas = smem + lo.atlas;
  uint16_t* sinfo = reinterpret_cast<uint16_t*>(smem + lo.sinfo);  // sprite | (player+1) << 8
  uint16_t* rinfo = reinterpret_cast<uint16_t*>(smem + lo.rinfo);  // remapped sprite | flags << 8
  uint16_t* slot = reinterpret_cast<uint16_t*>(smem + lo.slot);    // atlas image of (sprite, facing)
  uint8_t* wlds = smem + lo.world;                                 // [wpb][grid_pad + 64]
  const int wstride = t.grid_pad + kHeadBytes;
  uint32_t* offtab = reinterpret_cast<uint32_t*>(smem + lo.offtab);

  const int VW = t.vl + t.vr + 1, VH = t.vf + t.vb + 1;
  const int row_cells = kWorldView ? W : VW;
  const int strip_rows = kWorldView ? H : VH;   // strips per image
  const uint32_t row_bytes = (uint32_t)row_cells * 24u;
  const int lane = tid & 63, wave = tid >> 6;
  const int R = 64 / row_cells;                 // strips per wave pass
  const int ncell = R * row_cells;
  const int sr = (int)fast_div((uint32_t)lane, (uint32_t)row_cells, 1.0f / (float)row_cells);
  const uint32_t cx = (uint32_t)(lane - sr * row_cells);

  const int w_first = blockIdx.x * wpb;
  int nw = num_worlds - w_first;
  if (nw > wpb) nw = wpb;

  // ---- prologue: everything this workgroup will read, into LDS
  {
    const uint4* src = reinterpret_cast<const uint4*>(t.atlas_compact);
    for (int i = tid; i < t.n_images * 16; i += kThreads) {
      const int img = i >> 4, q = i & 15;
      reinterpret_cast<uint4*>(atlas + img * kSpriteStride)[q] = src[i];
    }
    for (int s = tid; s < 256; s += kThreads) {
      const int sp = s < t.nstates ? t.state_sprite[s] : -1;
      const int pl = s < t.nstates ? t.state_player[s] : -1;
      sinfo[s] = (uint16_t)((sp < 0 ? 0xff : sp) | ((pl + 1) << 8));
    }
    for (int i = tid; i < (P + 1) * t.nsprites; i += kThreads) {
      const int sp = t.view_sprite_map[i];
      rinfo[i] = (uint16_t)(sp | (t.sprite_flags8[sp] << 8));
    }
    for (int i = tid; i < t.nsprites * 4; i += kThreads) slot[i] = t.img_slot[i];
    if (tid < 64) offtab[tid] = (uint32_t)sr * 8u * row_bytes + cx * 24u;
    const int wvec = wstride >> 4;  // grid_pad and the 64-byte head are 16-byte multiples
    for (int i = tid; i < nw * wvec; i += kThreads) {
      const int lw = i / wvec, q = i - lw * wvec;
      const uint8_t* gw = state + (size_t)(w_first + lw) * t.world_stride;
      reinterpret_cast<uint4*>(wlds + lw * wstride)[q] = reinterpret_cast<const uint4*>(gw)[q];
    }
  }
  __syncthreads();

  const int strips_per_world = kWorldView ? H : P * VH;
  const uint32_t nstrips = (uint32_t)(nw * strips_per_world);
  uint8_t* out_block = out + (size_t)w_first * strips_per_world * 8 * row_bytes;
  CellRec* recs = reinterpret_cast<CellRec*>(smem + lo.recs) + wave * 64;
  const float rcp_rows = 1.0f / (float)strip_rows;
  const float rcp_p = 1.0f / (float)P;
  const int py = lane & 7;

  for (uint32_t s0 = (uint32_t)(wave * R); s0 < nstrips; s0 += (uint32_t)kWaves * R) {
    // ---- phase 1 (lane = cell): draw list of up to 10 entries of
    // (flags << 10 | atlas image), bottom -> top, restarted at every opaque
    // sprite.  An entry is never 0 (images are numbered from 1).
    if (ablate & 8) {
      CellRec r; r.la = (sr < R && s0 + sr < nstrips) ? (uint64_t)((1u << 10) | 5u) : kNoCell; r.lb = 0;
      recs[lane] = r;
    } else {
      const uint32_t strip = s0 + sr;
      const bool live = sr < R && strip < nstrips;
      const uint32_t sidx = live ? strip : 0u;
      const uint32_t img = fast_div(sidx, (uint32_t)strip_rows, rcp_rows);  // local world, or world*P + viewer
      const uint32_t cy = sidx - img * strip_rows;
      uint32_t lw = img, viewer = P, vo = 0;
      if (!kWorldView) {
        lw = fast_div(img, (uint32_t)P, rcp_p);
        viewer = img - lw * P;
      }
      const uint8_t* grid = wlds + lw * wstride;
      const uint8_t* head = grid + t.grid_pad;  // ax[16] ay[16] aori[16] aalive[16]
      int cell;
      if (kWorldView) {
        cell = (int)(cy * W + cx);
      } else {
        cell = -1;
        if (head[48 + viewer]) {  // A6: an off-grid viewer sees only OutOfBounds
          vo = head[32 + viewer];
          const int dx = (int)cx - t.vl, dy = (int)cy - t.vf;  // right, down in view frame
          int ax, ay;
          switch (vo) {
            case 0: ax = dx; ay = dy; break;
            case 1: ax = -dy; ay = dx; break;
            case 2: ax = -dx; ay = -dy; break;
            default: ax = dy; ay = -dx; break;
          }
          int x = head[viewer] + ax, y = head[16 + viewer] + ay;
          if (t.topology == 1) {
            x = ((x % W) + W) % W; y = ((y % H) + H) % H;
            cell = y * W + x;
          } else if (x >= 0 && x < W && y >= 0 && y < H) {
            cell = y * W + x;
          }
        }
      }
      const uint16_t* rm = rinfo + viewer * t.nsprites;
      const int ld = cell >= 0 ? cell : 0;
      // batches of independent LDS reads (state -> sprite -> remap/flags ->
      // atlas image), all unconditional so that they can be kept in flight
      uint32_t ent[kMaxLayers];
#pragma unroll
      for (int l = 0; l < kMaxLayers; ++l) ent[l] = grid[(l < L ? l : L - 1) * HW + ld];
#pragma unroll
      for (int l = 0; l < kMaxLayers; ++l) ent[l] = sinfo[ent[l]];
      uint32_t flg[kMaxLayers];
#pragma unroll
      for (int l = 0; l < kMaxLayers; ++l) {
        const uint32_t si = ent[l];
        const uint32_t sp = si & 255u;
        const uint32_t pl = si >> 8;
        const uint32_t ori = pl ? head[32 + pl - 1] : 0u;   // avatar cells only
        const uint32_t r = rm[sp == 255u ? 0u : sp];
        flg[l] = (sp == 255u || l >= L) ? 0xffffu : (r >> 8);
        ent[l] = ((r & 255u) << 2) | ((ori - vo) & 3u);
      }
#pragma unroll
      for (int l = 0; l < kMaxLayers; ++l) ent[l] = slot[ent[l]];
      uint64_t list0 = 0, list1 = 0;
      int n = 0;
      if (cell == -1) {
        const uint32_t r = rm[0];  // OutOfBounds sprite, facing north
        list0 = ((uint64_t)(r >> 8) << 10) | slot[(r & 255u) << 2];
      } else {
#pragma unroll
        for (int l = 0; l < kMaxLayers; ++l) {
          if (flg[l] == 0xffffu) continue;
          const uint64_t e = ((uint64_t)flg[l] << 10) | ent[l];
          if (flg[l] & FLAG_OPAQUE) { n = 0; list0 = 0; list1 = 0; }
          if (n < 5) list0 |= e << (12 * n); else if (n < 10) list1 |= e << (12 * (n - 5));
          ++n;
        }
      }
      CellRec r;
      r.la = live ? list0 : kNoCell;
      r.lb = list1;
      recs[lane] = r;
    }

    // ---- phase 2 (8 lanes per cell, one per pixel row)
    uint8_t* span = out_block + (size_t)s0 * 8 * row_bytes + (uint32_t)py * row_bytes;
    for (int g = 0; g * 8 < ncell; ++g) {
      const int c = g * 8 + (lane >> 3);
      if (c >= ncell) continue;
      const CellRec r = recs[c];
      if (r.la == kNoCell) continue;
      uint64_t la = r.la;
      uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (!(ablate & 2)) {
        for (int k = 0; k < 10; ++k) {
          const uint32_t e = (uint32_t)la & 4095u;
          if (e == 0) break;
          la >>= 12;
          if (k == 4) la = r.lb;
          const uint8_t* row = atlas + (e & 1023u) * kSpriteStride + py * 32;
          const uint32_t flags = e >> 10;
          if (flags & FLAG_OPAQUE) blend_row<0>(acc, row);
          else if (flags & FLAG_PARTIAL) blend_row<2>(acc, row);
          else blend_row<1>(acc, row);
        }
      }
      if (!(ablate & 1) || acc[0] == 0x12345678u) store_row(span + offtab[c], acc);
    }
  }
}

}  // namespace

int render_lds_bytes(const DevTables& t, int wpb) { return render_lds_layout(t, wpb).total; }

void launch_render(const DevTables& t, const uint8_t* state, uint8_t* out,
                   int num_worlds, bool world_view, int wpb, hipStream_t stream) {
  static const int ablate = getenv("MP_RENDER_ABLATE") ? atoi(getenv("MP_RENDER_ABLATE")) : 0;
  static const int wpb_env = getenv("MP_RENDER_WPB") ? atoi(getenv("MP_RENDER_WPB")) : 0;
  if (wpb_env > 0) wpb = wpb_env;
  const size_t lds = (size_t)render_lds_layout(t, wpb).total;
  const int blocks = (num_worlds + wpb - 1) / wpb;
  if (world_view)
    hipLaunchKernelGGL(k_render<true>, dim3(blocks), dim3(kThreads), lds, stream, t,
                       state, out, num_worlds, wpb, ablate);
  else
    hipLaunchKernelGGL(k_render<false>, dim3(blocks), dim3(kThreads), lds, stream, t,
                       state, out, num_worlds, wpb, ablate);
}
