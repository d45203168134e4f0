// Does the write rate of "every workgroup streams through its own multi-MB range"
// depend on WHERE the buffer lies?  (It does: the same frame launch runs 269 - 355 us
// on buffers of one process, while a plain fill runs 220 us on all of them.)
//
//   hipcc --offload-arch=gfx950 -O3 -o store_placement store_placement.hip && ./store_placement
//
// Six buffers of commons_harvest's per-agent output (4096 x 16 x 23,232 B), each
// written by: a grid-stride fill; persistent workgroups (one per CU, 12 storing
// waves, spans of 10,560 B = one pass of the frame kernel) over CONTIGUOUS ranges
// (18 worlds each, the frame kernel's assignment); the same with the ranges
// INTERLEAVED in chunks of one batch (3 worlds), one view, one span.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr uint32_t kSpan = 10560;                       // 5 strips x 8 rows x 264 B
constexpr uint64_t kBytes = 4096ull * 16 * 23232;       // 1.52 GB
constexpr uint32_t kSpans = (uint32_t)(kBytes / kSpan); // 144,179.2 -> floor
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ inline void write_span(uint8_t* span, int lane, uint32_t tag) {
#pragma unroll
  for (int it = 0; it < 11; ++it) {
    const uint32_t off = (uint32_t)(it * 64 + lane) * 16u;
    if (off < kSpan)
      asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(off), "v"(u32x4{tag, off, 2u, 3u}), "s"(span));
  }
}

// chunk = spans per chunk; workgroup g owns chunks g, g + G, g + 2G ...; chunk >=
// the group's whole share = contiguous ranges
__global__ void k_persistent(uint8_t* out, uint32_t chunk, uint32_t chunks_per_group) {
  __shared__ uint32_t next;
  if (threadIdx.x == 0) next = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t n = chunk * chunks_per_group;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&next, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (t >= n) break;
    const uint32_t c = t / chunk, i = t - c * chunk;
    const uint64_t s_idx = ((uint64_t)c * gridDim.x + blockIdx.x) * chunk + i;
    if (s_idx >= kSpans) continue;
    uint8_t* span = out + s_idx * kSpan;
    const uint64_t sp = reinterpret_cast<uint64_t>(span);
    span = reinterpret_cast<uint8_t*>(
        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sp));
    write_span(span, lane, t);
  }
}

__global__ void k_stride(uint4* out, uint64_t nvec) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nvec;
       i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

static hipEvent_t ea, eb;
template <class F>
static float best_of(F f, int reps = 6) {
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(ea, 0); f(); hipEventRecord(eb, 0); hipEventSynchronize(eb);
    float ms; hipEventElapsedTime(&ms, ea, eb);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  const int nbuf = 6;
  uint8_t* buf[nbuf];
  for (int b = 0; b < nbuf; ++b) CK(hipMalloc((void**)&buf[b], kBytes));
  const int G = getenv("GROUPS") ? atoi(getenv("GROUPS")) : 228, WAVES = 12;
  const uint32_t per_group = (kSpans + G - 1) / G;
  struct V { const char* name; uint32_t chunk; } vs[] = {
      {"contiguous range per workgroup", per_group}, {"interleaved by batch (3 worlds, 106 spans)", 106},
      {"interleaved by view (2.2 spans -> 2)", 2}, {"interleaved by span", 1}, {"interleaved by 11 spans", 11},
      {"interleaved by 35 spans (1 world)", 35}};
  printf("| buffer | grid-stride fill 16384 x 1024 |");
  for (auto& v : vs) printf(" %s |", v.name);
  printf("\n|---|---:|"); for (auto& v : vs) { (void)v; printf("---:|"); } printf("\n");
  for (int rep = 0; rep < 2; ++rep)
  for (int b = 0; b < nbuf; ++b) {
    const float tf = best_of([&] { hipLaunchKernelGGL(k_stride, dim3(16384), dim3(1024), 0, 0, (uint4*)buf[b], kBytes / 16); });
    printf("| %p | %.0f us %.2f TB/s |", (void*)buf[b], tf, kBytes / tf / 1e6);
    for (auto& v : vs) {
      const uint32_t cpg = (per_group + v.chunk - 1) / v.chunk;
      const float t = best_of([&] { hipLaunchKernelGGL(k_persistent, dim3(G), dim3(WAVES * 64), 0, 0, buf[b], v.chunk, cpg); });
      printf(" %.0f us %.2f |", t, kBytes / t / 1e6);
    }
    printf("\n"); fflush(stdout);
  }
  return 0;
}
