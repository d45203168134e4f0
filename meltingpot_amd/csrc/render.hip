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
// Execution shape: HBM-write bound.  A 256-thread workgroup owns one world at a
// time (grid-strided so the sprite atlas is staged into LDS once per
// workgroup): the world's grid planes are streamed into LDS, one pass builds a
// per-cell draw list (bottom->top sprites starting at the topmost fully opaque
// one, so hidden layers cost nothing), then every lane composites one 8-pixel
// sprite row (24 B) per item from LDS and stores it straight into the caller's
// tensor; consecutive lanes write consecutive 24-byte chunks of one pixel row.
#include "mp_common.h"

namespace {

struct RenderLds {
  // byte offsets into dynamic LDS
  int atlas, grid, tail, remap, ssprite, splayer, opaque, dl, dn, total;
};

__host__ __device__ inline RenderLds render_lds_layout(const DevTables& t) {
  RenderLds r;
  int off = 0;
  r.atlas = off; off += t.nsprites * 4 * 64 * 4;
  r.grid = off; off += t.grid_pad;
  r.tail = off; off += (int)sizeof(WorldTail);
  r.remap = off; off += ((t.P + 1) * t.nsprites + 15) & ~15;
  r.ssprite = off; off += 256;
  r.splayer = off; off += 256;
  r.opaque = off; off += 256;
  r.dl = off; off += ((t.H * t.W * t.L * 2) + 15) & ~15;
  r.dn = off; off += ((t.H * t.W) + 15) & ~15;
  r.total = off;
  return r;
}

// out = (src*a + dst*(255-a) + 127) / 255 per channel (A7); x/255 computed as
// (x + 1 + (x >> 8)) >> 8, exact for x < 65535 (max here 65152).
__device__ inline uint32_t blend_px(uint32_t dst, uint32_t src) {
  const uint32_t a = src >> 24;
  if (a == 255u) return src & 0xffffffu;
  if (a == 0u) return dst;
  const uint32_t ia = 255u - a;
  uint32_t out = 0;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const uint32_t s = (src >> (8 * ch)) & 255u, d = (dst >> (8 * ch)) & 255u;
    const uint32_t x = s * a + d * ia + 127u;
    out |= ((x + 1u + (x >> 8)) >> 8) << (8 * ch);
  }
  return out;
}

struct Row8 { uint32_t px[8]; };

__device__ inline void blend_row(Row8& acc, const uint8_t* atlas, int sprite,
                                 int facing, int py) {
  const uint4* row = reinterpret_cast<const uint4*>(
      atlas + (((sprite * 4 + facing) * 8 + py) << 5));
  const uint4 a = row[0], b = row[1];
  acc.px[0] = blend_px(acc.px[0], a.x); acc.px[1] = blend_px(acc.px[1], a.y);
  acc.px[2] = blend_px(acc.px[2], a.z); acc.px[3] = blend_px(acc.px[3], a.w);
  acc.px[4] = blend_px(acc.px[4], b.x); acc.px[5] = blend_px(acc.px[5], b.y);
  acc.px[6] = blend_px(acc.px[6], b.z); acc.px[7] = blend_px(acc.px[7], b.w);
}

// 8 RGB pixels -> 24 bytes, three 8-byte stores (dst is 8-byte aligned).
__device__ inline void store_row(uint8_t* dst, const Row8& r) {
  const uint32_t w0 = r.px[0] | (r.px[1] << 24);
  const uint32_t w1 = (r.px[1] >> 8) | (r.px[2] << 16);
  const uint32_t w2 = (r.px[2] >> 16) | (r.px[3] << 8);
  const uint32_t w3 = r.px[4] | (r.px[5] << 24);
  const uint32_t w4 = (r.px[5] >> 8) | (r.px[6] << 16);
  const uint32_t w5 = (r.px[6] >> 16) | (r.px[7] << 8);
  uint2* d = reinterpret_cast<uint2*>(dst);
  d[0] = make_uint2(w0, w1);
  d[1] = make_uint2(w2, w3);
  d[2] = make_uint2(w4, w5);
}

template <bool kWorldView>
__global__ __launch_bounds__(256) void k_render(DevTables t,
                                                const uint8_t* __restrict__ state,
                                                uint8_t* __restrict__ out,
                                                int num_worlds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const RenderLds lo = render_lds_layout(t);
  const int tid = threadIdx.x;
  const int HW = t.H * t.W, L = t.L, P = t.P, W = t.W, H = t.H;
  uint8_t* atlas = smem + lo.atlas;
  uint8_t* grid = smem + lo.grid;
  const WorldTail* tail = reinterpret_cast<const WorldTail*>(smem + lo.tail);
  uint8_t* remap = smem + lo.remap;
  uint8_t* ssprite = smem + lo.ssprite;
  int8_t* splayer = reinterpret_cast<int8_t*>(smem + lo.splayer);
  uint8_t* opaque = smem + lo.opaque;
  uint16_t* dl = reinterpret_cast<uint16_t*>(smem + lo.dl);
  uint8_t* dn = smem + lo.dn;

  // ---- once per workgroup: sprite atlas + lookup tables into LDS
  {
    const int nvec = t.nsprites * 64;  // 1 KiB per sprite = 64 uint4
    const uint4* src = reinterpret_cast<const uint4*>(t.sprite_rgba);
    for (int i = tid; i < nvec; i += 256) reinterpret_cast<uint4*>(atlas)[i] = src[i];
    for (int i = tid; i < (P + 1) * t.nsprites; i += 256)
      remap[i] = (uint8_t)t.view_sprite_map[i];
    for (int s = tid; s < 256; s += 256) {
      const int sp = s < t.nstates ? t.state_sprite[s] : -1;
      ssprite[s] = sp < 0 ? 0xff : (uint8_t)sp;
      splayer[s] = s < t.nstates ? t.state_player[s] : (int8_t)-1;
      opaque[s] = s < t.nsprites ? t.sprite_opaque[s] : 0;
    }
  }

  for (int w = blockIdx.x; w < num_worlds; w += gridDim.x) {
    __syncthreads();  // previous world's pixel phase is done with LDS
    const uint8_t* gw = state + (size_t)w * t.world_stride;
    {
      const int gvec = t.grid_pad >> 4;
      for (int i = tid; i < gvec; i += 256)
        reinterpret_cast<uint4*>(grid)[i] = reinterpret_cast<const uint4*>(gw)[i];
      const int tvec = (int)sizeof(WorldTail) >> 4;
      if (tid < tvec)
        reinterpret_cast<uint4*>(smem + lo.tail)[tid] =
            reinterpret_cast<const uint4*>(gw + t.grid_pad)[tid];
    }
    __syncthreads();
    // ---- per-cell draw list: (sprite << 2 | piece orientation), bottom -> top,
    // starting at the topmost sprite that is opaque under every spriteMap.
    for (int cell = tid; cell < HW; cell += 256) {
      int l0 = 0;
      for (int l = L - 1; l >= 0; --l) {
        const int s = grid[l * HW + cell];
        if (s == 0) continue;
        const int sp = ssprite[s];
        if (sp != 0xff && opaque[sp]) { l0 = l; break; }
      }
      int n = 0;
      for (int l = l0; l < L; ++l) {
        const int s = grid[l * HW + cell];
        if (s == 0) continue;
        const int sp = ssprite[s];
        if (sp == 0xff) continue;
        const int pl = splayer[s];
        const int ori = pl >= 0 ? tail->aori[pl] : 0;
        dl[cell * L + n++] = (uint16_t)((sp << 2) | ori);
      }
      dn[cell] = (uint8_t)n;
    }
    __syncthreads();

    if (kWorldView) {
      // "WORLD.RGB": viewer row P of the sprite map, facing north.
      const uint8_t* rm = remap + P * t.nsprites;
      const int items = H * 8 * W;
      uint8_t* ow = out + (size_t)w * items * 24;
      for (int i = tid; i < items; i += 256) {
        const int r = i / W, cx = i - r * W;
        const int cell = (r >> 3) * W + cx, py = r & 7;
        Row8 acc = {{0, 0, 0, 0, 0, 0, 0, 0}};
        const int n = dn[cell];
        for (int k = 0; k < n; ++k) {
          const int e = dl[cell * L + k];
          blend_row(acc, atlas, rm[e >> 2], e & 3, py);
        }
        store_row(ow + (size_t)i * 24, acc);
      }
    } else {
      // "N.RGB": egocentric window, rotated so that the avatar faces up.
      const int VW = t.vl + t.vr + 1, VH = t.vf + t.vb + 1;
      const int per_view = VH * 8 * VW;
      const int items = P * per_view;
      uint8_t* ow = out + (size_t)w * items * 24;
      for (int i = tid; i < items; i += 256) {
        const int v = i / per_view, rem = i - v * per_view;
        const int r = rem / VW, vx = rem - r * VW;
        const int vy = r >> 3, py = r & 7;
        const uint8_t* rm = remap + v * t.nsprites;
        Row8 acc = {{0, 0, 0, 0, 0, 0, 0, 0}};
        int cell = -1, vo = 0;
        if (tail->aalive[v]) {  // A6: an off-grid viewer sees only OutOfBounds
          vo = tail->aori[v];
          const int dx = vx - t.vl, dy = vy - t.vf;  // right, down in view frame
          int ax, ay;
          switch (vo) {
            case 0: ax = dx; ay = dy; break;
            case 1: ax = -dy; ay = dx; break;
            case 2: ax = -dx; ay = -dy; break;
            default: ax = dy; ay = -dx; break;
          }
          int x = tail->ax[v] + ax, y = tail->ay[v] + ay;
          if (t.topology == 1) {
            x = ((x % W) + W) % W; y = ((y % H) + H) % H;
            cell = y * W + x;
          } else if (x >= 0 && x < W && y >= 0 && y < H) {
            cell = y * W + x;
          }
        }
        if (cell < 0) {
          blend_row(acc, atlas, rm[0], 0, py);  // OutOfBounds sprite
        } else {
          const int n = dn[cell];
          for (int k = 0; k < n; ++k) {
            const int e = dl[cell * L + k];
            blend_row(acc, atlas, rm[e >> 2], ((e & 3) - vo) & 3, py);
          }
        }
        store_row(ow + (size_t)i * 24, acc);
      }
    }
  }
}

}  // namespace

int render_lds_bytes(const DevTables& t) { return render_lds_layout(t).total; }

void launch_render(const DevTables& t, const uint8_t* state, uint8_t* out,
                   int num_worlds, bool world_view, int num_blocks,
                   hipStream_t stream) {
  const size_t lds = (size_t)render_lds_layout(t).total;
  if (world_view)
    hipLaunchKernelGGL(k_render<true>, dim3(num_blocks), dim3(256), lds, stream, t,
                       state, out, num_worlds);
  else
    hipLaunchKernelGGL(k_render<false>, dim3(num_blocks), dim3(256), lds, stream,
                       t, state, out, num_worlds);
}
