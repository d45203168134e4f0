// At which granularity is memory spread over what lies behind the fabric (HBM stacks, IODs)?
//
//   hipcc --offload-arch=gfx950 -O3 -o class_bw class_bw.hip && ./class_bw [GiB] [xcd_mask_hex]
//
// The WHOLE chip (or the XCDs of `xcd_mask`) writes only one CLASS of a physically contiguous
// extent: the blocks of 2^s bytes with (offset >> s) % m == c, as a chip-wide front of 16-byte
// lane-contiguous vectors (the placement-insensitive order of profiles/r04_write_fronts.md).
// If the resource that limits a saturated write — a stack, an IOD's memory side, a fabric
// link — is selected by address bits [s, s + log2 m), a class hits 1/m of it and the rate
// drops towards 1/m of the full rate; where the rate stays flat the bits are hashed away or
// interleaved finer.  One table per m: rows s, columns c, TB/s.  (profiles/r05_alloc_method.md)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_class(uint8_t* buf, uint64_t bytes, uint32_t xcd_mask, uint32_t rank_of_xcd,
                                               int nxcd, int m, int cls, int shift) {
  const uint32_t xcd = blockIdx.x & 7;
  if (!((xcd_mask >> xcd) & 1u)) return;
  // this workgroup's index among the participating ones (XCD-major inside a round of 8)
  const uint32_t w = (blockIdx.x >> 3) * (uint32_t)nxcd + ((rank_of_xcd >> (4 * xcd)) & 15u);
  const uint32_t W = (gridDim.x >> 3) * (uint32_t)nxcd;
  const uint64_t vec_per_blk = (1ull << shift) / 16;
  const uint64_t nblk = bytes >> shift;
  const uint64_t nvec = (nblk / (uint64_t)m) * vec_per_blk;     // vectors of this class
  for (uint64_t v = (uint64_t)w * blockDim.x + threadIdx.x; v < nvec; v += (uint64_t)W * blockDim.x) {
    const uint64_t bi = v / vec_per_blk, j = v - bi * vec_per_blk;
    uint8_t* p = buf + (((bi * (uint64_t)m + (uint64_t)cls) << shift) + j * 16);
    asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(u32x4{1u, 2u, 3u, 4u}));
  }
}

int main(int argc, char** argv) {
  const uint64_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4;
  const uint32_t mask = argc > 2 ? (uint32_t)strtoul(argv[2], nullptr, 16) : 0xffu;
  const uint64_t bytes = gib << 30;
  hipEvent_t ea, eb;
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  uint8_t* buf = nullptr;
  const char* what = "physically contiguous extent";
  if (hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocContiguous) != hipSuccess || !buf) {
    (void)hipGetLastError();
    CK(hipMalloc((void**)&buf, bytes));
    what = "hipMalloc (no contiguous extent of that size)";
  }
  CK(hipMemset(buf, 1, bytes));
  int nxcd = 0; uint32_t ranks = 0;
  for (int x = 0; x < 8; ++x) if ((mask >> x) & 1u) { ranks |= (uint32_t)nxcd << (4 * x); ++nxcd; }
  printf("# class_bw: %llu GiB, %s at %p, XCD mask 0x%02x (%d XCDs)\n\n", (unsigned long long)gib, what, (void*)buf, mask, nxcd);
  auto run = [&](int m, int cls, int shift) {
    std::vector<float> t;
    for (int r = 0; r < 4; ++r) {
      hipEventRecord(ea, 0);
      hipLaunchKernelGGL(k_class, dim3(256), dim3(512), 0, 0, buf, bytes, mask, ranks, nxcd, m, cls, shift);
      hipEventRecord(eb, 0); hipEventSynchronize(eb);
      float ms; hipEventElapsedTime(&ms, ea, eb);
      if (r) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double nb = (double)((bytes >> shift) / (uint64_t)m) * (double)(1ull << shift);
    return nb / (t[t.size() / 2] * 1e-3) / 1e12;
  };
  // clocks up
  for (int r = 0; r < 30; ++r) run(1, 0, 20);
  printf("everything (m = 1): %.2f TB/s\n\n", run(1, 0, 20));
  for (int m : {2, 8}) {
    printf("## m = %d classes: TB/s writing only the blocks with (offset >> s) %% %d == c\n\n| s | block |", m, m);
    for (int c = 0; c < m; ++c) printf(" c=%d |", c);
    printf(" min / max |\n|---|---|"); for (int c = 0; c < m; ++c) printf("---:|"); printf("---|\n");
    for (int shift = 8; (bytes >> shift) >= (uint64_t)(2 * m); ++shift) {
      double lo = 1e9, hi = 0;
      printf("| %d | %s |", shift, shift < 10 ? "<1K" : shift < 20 ? (std::to_string(1 << (shift - 10)) + "K").c_str()
                                                                    : (std::to_string(1 << (shift - 20)) + "M").c_str());
      for (int c = 0; c < m; ++c) { const double v = run(m, c, shift); lo = std::min(lo, v); hi = std::max(hi, v); printf(" %.2f |", v); }
      printf(" %.2f / %.2f |\n", lo, hi);
      fflush(stdout);
    }
    printf("\n");
  }
  // ---- locality: which XCDs write which class fastest?  (all-XCD tables above show bit 23
  // matters; here subsets of the XCDs write one value of one address bit)
  if (argc > 3) return 0;
  const uint32_t masks[] = {0x0f, 0xf0, 0x33, 0xcc, 0x55, 0xaa, 0x03, 0x0c, 0x30, 0xc0, 0x01, 0x02, 0x10, 0x80};
  printf("## subsets of the XCDs writing one value of ONE address bit (m = 2): TB/s\n\n| XCD mask |");
  const int bits[] = {21, 22, 23, 24, 25};
  for (int b : bits) printf(" bit %d = 0 | bit %d = 1 |", b, b);
  printf("\n|---|"); for (size_t i = 0; i < 2 * sizeof bits / sizeof bits[0]; ++i) printf("---:|"); printf("\n");
  for (uint32_t mk : masks) {
    int n = 0; uint32_t rk = 0;
    for (int x = 0; x < 8; ++x) if ((mk >> x) & 1u) { rk |= (uint32_t)n << (4 * x); ++n; }
    printf("| 0x%02x |", mk);
    for (int b : bits)
      for (int c = 0; c < 2; ++c) {
        std::vector<float> t;
        for (int r = 0; r < 4; ++r) {
          hipEventRecord(ea, 0);
          hipLaunchKernelGGL(k_class, dim3(256), dim3(512), 0, 0, buf, bytes, mk, rk, n, 2, c, b);
          hipEventRecord(eb, 0); hipEventSynchronize(eb);
          float ms; hipEventElapsedTime(&ms, ea, eb);
          if (r) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf(" %.2f |", (double)(bytes / 2) / (t[t.size() / 2] * 1e-3) / 1e12);
      }
    printf("\n"); fflush(stdout);
  }
  return 0;
}
