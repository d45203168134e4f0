// How long does a workgroup wait for its kernel arguments?  (profiles/r04_head.md: 1.5 - 1.9 us between a
// frame launch's first instruction and its first use of an argument.)
//   hipcc --offload-arch=gfx950 -O3 -o kernarg_latency kernarg_latency.hip && ./kernarg_latency
// A kernel with 1.3 KB of by-value arguments stamps wall_clock64() (100 MHz) at entry, after it has ONE
// argument dword (the first line), after one dword of every 64-byte line (all in flight at once), and
// after a dependent chain of four lines; per workgroup, wave 0.  Launched back to back like the engine's
// steps (the argument block is rewritten by the host for every launch).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { uint32_t w[320]; };   // 1280 bytes
__global__ __launch_bounds__(768) void k(Big a, uint32_t* out, uint32_t salt) {
  const uint64_t t0 = wall_clock64();
  typedef const uint32_t __attribute__((address_space(4))) KW;
  KW* ka = (KW*)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t x = ka[0];
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x));
  const uint64_t t1 = wall_clock64();
  uint32_t y = 0;
#pragma unroll
  for (int i = 16; i < 320; i += 16) y ^= ka[i];
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(y));
  const uint64_t t2 = wall_clock64();
  // a dependent chain: the next line's index comes out of the previous load (values are < 300)
  uint32_t z = ka[(x & 255u) + 1];
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(z));
  z = ka[(z & 255u) + 2];
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(z));
  const uint64_t t3 = wall_clock64();
  if (threadIdx.x == 0) {
    uint32_t* o = out + blockIdx.x * 4;
    o[0] = (uint32_t)(t1 - t0); o[1] = (uint32_t)(t2 - t1); o[2] = (uint32_t)(t3 - t2); o[3] = x ^ y ^ z ^ salt;
  }
}
// the same block read through a pointer to DEVICE memory that never changes between launches
__global__ __launch_bounds__(768) void k2(const uint32_t* __restrict__ blk, uint32_t* out, uint32_t salt) {
  const uint64_t t0 = wall_clock64();
  typedef const uint32_t __attribute__((address_space(4))) KW;
  KW* ka = (KW*)blk;
  uint32_t x = ka[0];
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x));
  const uint64_t t1 = wall_clock64();
  uint32_t y = 0;
#pragma unroll
  for (int i = 16; i < 320; i += 16) y ^= ka[i];
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(y));
  const uint64_t t2 = wall_clock64();
  if (threadIdx.x == 0) {
    uint32_t* o = out + blockIdx.x * 4;
    o[0] = (uint32_t)(t1 - t0); o[1] = (uint32_t)(t2 - t1); o[2] = 0; o[3] = x ^ y ^ salt;
  }
}
int main() {
  uint32_t* out; CK(hipMalloc((void**)&out, 256 * 16));
  Big a; for (int i = 0; i < 320; ++i) a.w[i] = (uint32_t)(i * 7 % 200);
  std::vector<uint32_t> h(256 * 4);
  for (int rep = 0; rep < 5; ++rep) {
    for (int i = 0; i < 20; ++i) { a.w[5] = (uint32_t)i; hipLaunchKernelGGL(k, dim3(256), dim3(768), 0, 0, a, out, (uint32_t)i); }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), out, 256 * 16, hipMemcpyDeviceToHost));
    for (int c = 0; c < 3; ++c) {
      std::vector<uint32_t> v; for (int g = 0; g < 256; ++g) v.push_back(h[g * 4 + c]);
      std::sort(v.begin(), v.end());
      printf("%s: min %.2f  median %.2f  90%% %.2f  max %.2f us   ", c == 0 ? "first dword" : c == 1 ? "all 20 lines" : "2 dependent", v[0] / 100.0, v[128] / 100.0, v[230] / 100.0, v[255] / 100.0);
    }
    printf("\n");
  }
  uint32_t* blk; CK(hipMalloc((void**)&blk, 1280)); CK(hipMemcpy(blk, a.w, 1280, hipMemcpyHostToDevice));
  printf("through a device pointer (pointer itself = first kernarg line):\n");
  for (int rep = 0; rep < 4; ++rep) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k2, dim3(256), dim3(768), 0, 0, (const uint32_t*)blk, out, (uint32_t)i);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), out, 256 * 16, hipMemcpyDeviceToHost));
    for (int c = 0; c < 2; ++c) {
      std::vector<uint32_t> v; for (int g = 0; g < 256; ++g) v.push_back(h[g * 4 + c]);
      std::sort(v.begin(), v.end());
      printf("%s: min %.2f  median %.2f  90%% %.2f  max %.2f us   ", c == 0 ? "first dword (pointer + load)" : "all 20 lines", v[0] / 100.0, v[128] / 100.0, v[230] / 100.0, v[255] / 100.0);
    }
    printf("\n");
  }
  return 0;
}
