// Do the 8 XCDs see a buffer's memory equally?  (profiles/r04_write_fronts.md: with the
// work split evenly, the XCDs finish a 495 MB write 58 ... 97 us after the start, in
// pairs — an IOD — and the pattern is a property of the BUFFER.)
//
//   hipcc --offload-arch=gfx950 -O3 -o xcd_affinity xcd_affinity.hip && ./xcd_affinity
//
// One XCD at a time (the workgroups with blockIdx % 8 == x; the others exit) writes one
// CLASS of a buffer's bytes: class j of granularity 2^s = the blocks of 2^s bytes with
// (address >> s) % 8 == j, or the j-th contiguous eighth.  An 8 x 8 table of GB/s per
// granularity: a diagonal (or any structure) says the address interleave over the
// memory behind the IODs is visible at that granularity; a flat table says it is not.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// shift < 0: contiguous eighths.  Each active workgroup takes blocks round-robin.
template <bool kRead>
__global__ __launch_bounds__(512) void k_class(uint8_t* buf, uint64_t bytes, int xcd, int cls, int shift,
                                               uint32_t* sink) {
  if ((int)(blockIdx.x & 7) != xcd) return;
  const uint32_t member = blockIdx.x >> 3, members = gridDim.x >> 3;
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  uint32_t acc = 0;
  if (shift < 0) {
    const uint64_t eighth = (bytes / 8) & ~1023ull;
    uint8_t* base = buf + eighth * cls;
    // 1 KiB per wave store, the workgroups interleaved at 8 KiB
    for (uint64_t off = (uint64_t)member * nthr * 16 + tid * 16; off + 16 <= eighth; off += (uint64_t)members * nthr * 16) {
      if (kRead) acc += reinterpret_cast<const uint4*>(base + off)->x;
      else asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(base + off), "v"(u32x4{1u, 2u, 3u, 4u}));
    }
  } else {
    const uint64_t blk = 1ull << shift;
    const uint64_t nblk = bytes >> shift;             // blocks in the buffer
    const uint64_t per_class = nblk / 8;
    // the address class is taken on the VIRTUAL address (equal to the physical one below
    // the fragment size: 2 MB for these allocations)
    const uint64_t a0 = (reinterpret_cast<uint64_t>(buf) >> shift) & 7;
    const uint32_t vec_per_blk = (uint32_t)(blk / 16);
    // thread t of the workgroup handles 16-byte vector (t % vec_per_blk) of block (t / vec_per_blk)
    const uint32_t blks_per_iter = nthr >= vec_per_blk ? nthr / vec_per_blk : 1;
    const uint32_t iters_per_blk = nthr >= vec_per_blk ? 1 : vec_per_blk / nthr;
    for (uint64_t i = (uint64_t)member * blks_per_iter; i < per_class; i += (uint64_t)members * blks_per_iter) {
      const uint64_t bi = i + (nthr >= vec_per_blk ? tid / vec_per_blk : 0);
      if (bi >= per_class) continue;
      // block index b with ((a0 + b) % 8) == cls
      const uint64_t b = bi * 8 + ((cls + 8 - a0) & 7);
      if (b >= nblk) continue;
      uint8_t* p = buf + (b << shift);
      for (uint32_t it = 0; it < iters_per_blk; ++it) {
        const uint32_t v = nthr >= vec_per_blk ? tid % vec_per_blk : it * nthr + tid;
        if (kRead) acc += reinterpret_cast<const uint4*>(p + v * 16)->x;
        else asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p + v * 16), "v"(u32x4{1u, 2u, 3u, 4u}));
      }
    }
  }
  if (kRead && acc == 0x12345678u) sink[0] = acc;
}

static hipEvent_t ea, eb;
template <class F>
static float med_us(F f, int reps) {
  std::vector<float> v;
  f();
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(ea, 0); f(); hipEventRecord(eb, 0); hipEventSynchronize(eb);
    float ms; hipEventElapsedTime(&ms, ea, eb); v.push_back(ms * 1e3f);
  }
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char** argv) {
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  const uint64_t bytes = 4096ull * 120960;
  const bool do_read = argc > 1 && atoi(argv[1]) == 1;
  uint32_t* sink; CK(hipMalloc((void**)&sink, 64));
  std::vector<uint8_t*> bufs; std::vector<std::string> names;
  for (int b = 0; b < 2; ++b) { uint8_t* p; CK(hipMalloc((void**)&p, bytes)); CK(hipMemset(p, 1, bytes)); bufs.push_back(p); names.push_back("malloc" + std::to_string(b)); }
  { uint8_t* p = nullptr; if (hipExtMallocWithFlags((void**)&p, bytes, hipDeviceMallocContiguous) == hipSuccess && p) { CK(hipMemset(p, 1, bytes)); bufs.push_back(p); names.push_back("contig"); } else (void)hipGetLastError(); }
  const int shifts[] = {-1, 8, 10, 12, 13, 16, 21};
  for (size_t bi = 0; bi < bufs.size(); ++bi) {
    printf("## %s (%p), %s, one XCD at a time (32 workgroups x 512 threads)\n\n", names[bi].c_str(), (void*)bufs[bi], do_read ? "reads" : "nt writes");
    for (int shift : shifts) {
      if (shift < 0) printf("contiguous eighths, GB/s (row = XCD, column = eighth)\n\n");
      else printf("blocks of %d B, class = (address >> %d) %% 8, GB/s (row = XCD, column = class)\n\n", 1 << shift, shift);
      printf("| XCD | 0 | 1 | 2 | 3 | 4 | 5 | 6 | 7 |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|\n");
      for (int x = 0; x < 8; ++x) {
        printf("| %d |", x);
        for (int j = 0; j < 8; ++j) {
          const float us = med_us([&] {
            if (do_read) hipLaunchKernelGGL(k_class<true>, dim3(256), dim3(512), 0, 0, bufs[bi], bytes, x, j, shift, sink);
            else hipLaunchKernelGGL(k_class<false>, dim3(256), dim3(512), 0, 0, bufs[bi], bytes, x, j, shift, sink);
          }, 3);
          printf(" %.0f |", (double)(bytes / 8) / us / 1e3);
        }
        printf("\n");
      }
      printf("\n"); fflush(stdout);
    }
  }
  return 0;
}
