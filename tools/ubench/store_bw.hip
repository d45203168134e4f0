// Micro-benchmark: HBM write bandwidth for the renderer's store patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void fill16(uint4* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void fill16nt(uint4* out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    __builtin_nontemporal_store((uint32_t)i, &out[i].x);
    __builtin_nontemporal_store(1u, &out[i].y);
    __builtin_nontemporal_store(2u, &out[i].z);
    __builtin_nontemporal_store(3u, &out[i].w);
  }
}
__global__ void fill8x3(uint2* out, size_t nitems) {  // item = 24 B
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nitems; i += (size_t)gridDim.x * blockDim.x) {
    uint2* d = out + i * 3;
    d[0] = make_uint2((uint32_t)i, 1); d[1] = make_uint2(2, 3); d[2] = make_uint2(4, 5);
  }
}
__global__ void fill4x6(uint32_t* out, size_t nitems) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nitems; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t* d = out + i * 6;
    for (int k = 0; k < 6; ++k) d[k] = (uint32_t)i + k;
  }
}
// per block: contiguous chunk per wave, written as 16 B per lane (what LDS staging would give)
__global__ void fill16_blockchunk(uint4* out, size_t n, int per_block) {
  for (size_t base = (size_t)blockIdx.x * per_block; base < n; base += (size_t)gridDim.x * per_block)
    for (int j = threadIdx.x; j < per_block && base + j < n; j += blockDim.x)
      out[base + j] = make_uint4((uint32_t)j, 1, 2, 3);
}

int main() {
  const size_t bytes = (size_t)4096 * 120960;
  void* buf; CK(hipMalloc(&buf, bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
    for (int variant = 0; variant < 5; ++variant) {
      float best = 1e9;
      for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(a));
        switch (variant) {
          case 0: fill16<<<blocks, 256>>>((uint4*)buf, bytes / 16); break;
          case 1: fill16nt<<<blocks, 256>>>((uint4*)buf, bytes / 16); break;
          case 2: fill8x3<<<blocks, 256>>>((uint2*)buf, bytes / 24); break;
          case 3: fill4x6<<<blocks, 256>>>((uint32_t*)buf, bytes / 24); break;
          case 4: fill16_blockchunk<<<blocks, 256>>>((uint4*)buf, bytes / 16, 7560); break;
        }
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;
      }
      const char* names[] = {"fill16", "fill16nt", "fill8x3", "fill4x6", "fill16_chunk"};
      printf("blocks=%5d %-14s %.1f us  %.0f GB/s\n", blocks, names[variant], best * 1e3, bytes / (best * 1e-3) / 1e9);
    }
  }
  return 0;
}
