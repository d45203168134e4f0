// Micro-benchmark 2: the renderer's exact strip store pattern vs alternatives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int W = 30, R = 2;               // cells per strip, strips per wave
constexpr int ROWB = W * 24;               // 720

// V0: lane = (strip, cx); 8 rows x 3 x 8-byte stores
__global__ void v0(uint8_t* out, uint32_t total_strips, uint32_t per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sr = lane / W, cx = lane % W;
  uint32_t begin = blockIdx.x * per_block, end = min(begin + per_block, total_strips);
  for (uint32_t s0 = begin + wave * R; s0 < end; s0 += 4 * R) {
    uint32_t strip = s0 + sr;
    if (sr >= R || strip >= end) continue;
    uint8_t* dst = out + (size_t)strip * 8 * ROWB + cx * 24;
#pragma unroll
    for (int py = 0; py < 8; ++py) {
      uint2* d = (uint2*)(dst + py * ROWB);
      d[0] = make_uint2(strip, py); d[1] = make_uint2(cx, 1); d[2] = make_uint2(2, 3);
    }
  }
}
// V1: 16+8 byte stores depending on parity
__global__ void v1(uint8_t* out, uint32_t total_strips, uint32_t per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sr = lane / W, cx = lane % W;
  uint32_t begin = blockIdx.x * per_block, end = min(begin + per_block, total_strips);
  for (uint32_t s0 = begin + wave * R; s0 < end; s0 += 4 * R) {
    uint32_t strip = s0 + sr;
    if (sr >= R || strip >= end) continue;
    uint8_t* dst = out + (size_t)strip * 8 * ROWB + cx * 24;
#pragma unroll
    for (int py = 0; py < 8; ++py) {
      uint8_t* d = dst + py * ROWB;
      if (cx & 1) { *(uint2*)d = make_uint2(strip, py); *(uint4*)(d + 8) = make_uint4(cx, 1, 2, 3); }
      else { *(uint4*)d = make_uint4(strip, py, cx, 1); *(uint2*)(d + 16) = make_uint2(2, 3); }
    }
  }
}
// V2: wave writes its contiguous span (R strips = R*8*ROWB bytes) as 16 B per lane
__global__ void v2(uint8_t* out, uint32_t total_strips, uint32_t per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t begin = blockIdx.x * per_block, end = min(begin + per_block, total_strips);
  for (uint32_t s0 = begin + wave * R; s0 < end; s0 += 4 * R) {
    uint32_t ns = min((uint32_t)R, end - s0);
    uint4* dst = (uint4*)(out + (size_t)s0 * 8 * ROWB);
    const int nvec = ns * 8 * ROWB / 16;
    for (int i = lane; i < nvec; i += 64) dst[i] = make_uint4(s0, i, 2, 3);
  }
}
// V3: like V0 but one strip per wave pass using all 8 rows: lane = (py, chunk of 8 cells?) -> lane handles
// row py (8 lanes per row?) : lane = py*8 + j, j in 0..7 writes cells j, j+8, j+16, j+24(<30): contiguous 24 B each
__global__ void v3(uint8_t* out, uint32_t total_strips, uint32_t per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int py = lane >> 3, j = lane & 7;
  uint32_t begin = blockIdx.x * per_block, end = min(begin + per_block, total_strips);
  for (uint32_t strip = begin + wave; strip < end; strip += 4) {
    uint8_t* row = out + (size_t)strip * 8 * ROWB + py * ROWB;
    for (int cx = j; cx < W; cx += 8) {
      uint2* d = (uint2*)(row + cx * 24);
      d[0] = make_uint2(strip, py); d[1] = make_uint2(cx, 1); d[2] = make_uint2(2, 3);
    }
  }
}

// V4: like V3 but with a delay between the 4 sub-iterations of a row (what LDS round trips do in the renderer)
template <int kSleep>
__global__ void v4(uint8_t* out, uint32_t total_strips, uint32_t per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int py = lane >> 3, j = lane & 7;
  uint32_t begin = blockIdx.x * per_block, end = min(begin + per_block, total_strips);
  for (uint32_t strip = begin + wave; strip < end; strip += 4) {
    uint8_t* row = out + (size_t)strip * 8 * ROWB + py * ROWB;
    for (int cx = j; cx < W; cx += 8) {
      uint2* d = (uint2*)(row + cx * 24);
      d[0] = make_uint2(strip, py); d[1] = make_uint2(cx, 1); d[2] = make_uint2(2, 3);
      for (int k = 0; k < kSleep; ++k) __builtin_amdgcn_s_sleep(8);  // 8*64 cycles
    }
  }
}
// V5: lane = (cell c of 8, row py) sub-passes (the renderer's phase-2 mapping), optional delay
template <int kSleep>
__global__ void v5(uint8_t* out, uint32_t total_strips, uint32_t per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int py = lane & 7;
  uint32_t begin = blockIdx.x * per_block, end = min(begin + per_block, total_strips);
  for (uint32_t s0 = begin + wave * R; s0 < end; s0 += 4 * R) {
    uint8_t* span = out + (size_t)s0 * 8 * ROWB + py * ROWB;
    for (int g = 0; g * 8 < R * W; ++g) {
      const int c = g * 8 + (lane >> 3);
      if (c >= R * W || s0 + c / W >= end) continue;
      uint2* d = (uint2*)(span + (c / W) * 8 * ROWB + (c % W) * 24);
      d[0] = make_uint2(s0, py); d[1] = make_uint2(c, 1); d[2] = make_uint2(2, 3);
      for (int k = 0; k < kSleep; ++k) __builtin_amdgcn_s_sleep(8);
    }
  }
}

int main() {
  const uint32_t strips = 4096 * 21;
  const size_t bytes = (size_t)strips * 8 * ROWB;
  void* buf; CK(hipMalloc(&buf, bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int blocks : {1280, 4096}) {
    uint32_t per_block = (strips + blocks - 1) / blocks; per_block = (per_block + 7) / 8 * 8;
    uint32_t nb = (strips + per_block - 1) / per_block;
    for (int variant = 0; variant < 8; ++variant) {
      float best = 1e9;
      for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(a));
        switch (variant) {
          case 0: v0<<<nb, 256>>>((uint8_t*)buf, strips, per_block); break;
          case 1: v1<<<nb, 256>>>((uint8_t*)buf, strips, per_block); break;
          case 2: v2<<<nb, 256>>>((uint8_t*)buf, strips, per_block); break;
          case 3: v3<<<nb, 256>>>((uint8_t*)buf, strips, per_block); break;
          case 4: v4<1><<<nb, 256>>>((uint8_t*)buf, strips, per_block); break;
          case 5: v5<0><<<nb, 256>>>((uint8_t*)buf, strips, per_block); break;
          case 6: v5<1><<<nb, 256>>>((uint8_t*)buf, strips, per_block); break;
          case 7: v5<4><<<nb, 256>>>((uint8_t*)buf, strips, per_block); break;
        }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;
      }
      const char* names[] = {"v0 strip 8x3x8B", "v1 strip 16+8", "v2 span 16B/lane", "v3 row-major lanes", "v4 v3+sleep512", "v5 cell-major sub", "v5+sleep512", "v5+sleep2048"};
      printf("blocks=%5u %-20s %.1f us  %.0f GB/s\n", nb, names[variant], best * 1e3, bytes / (best * 1e-3) / 1e9);
    }
  }
  return 0;
}
