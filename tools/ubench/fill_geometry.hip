// How many concurrent streams should write a view?  (round 6, profiles/r06_fill_geometry.md)
//
//   hipcc --offload-arch=gfx950 -O3 -o fill_geometry fill_geometry.hip && ./fill_geometry [shape]
//
// mp_box_fill (bench.py's `box_fill`) showed, on one box, the bare store loop in the frame
// launch's order take 97 us for clean_up's WORLD.RGB with the headline plan's geometry (256
// workgroups x 8 storing waves) and 77 us for the same bytes with the both-views plan's (228 x
// 6) — on different buffers.  This takes the buffer out: the same buffers, the product order
// (every workgroup a contiguous range of whole worlds, whole spans per wave from an LDS
// ticket counter, 16-byte lane-contiguous nt stores) and the 4 KiB chip-wide front, over
// workgroups x storing waves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void k_fill(uint8_t* out, uint64_t bytes, uint64_t own, uint32_t span, int order) {
  __shared__ uint32_t next;
  if (threadIdx.x == 0) next = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t g = blockIdx.x, G = gridDim.x;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&next, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    uint64_t begin, end;
    if (order == 0) {
      uint64_t lim = (g + 1) * own; if (lim > bytes) lim = bytes;
      begin = g * own + (uint64_t)t * span;
      if (begin >= lim) break;
      end = begin + span < lim ? begin + span : lim;
    } else {
      begin = ((uint64_t)t * G + g) * span;
      if (begin >= bytes) break;
      end = begin + span < bytes ? begin + span : bytes;
    }
    const uint64_t sp = reinterpret_cast<uint64_t>(out + begin);
    uint8_t* base = reinterpret_cast<uint8_t*>(
        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sp));
    const uint32_t n = (uint32_t)(end - begin);
    for (uint32_t off = lane * 16u; off < n; off += 1024u)
      asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(off), "v"(u32x4{t, off, 2u, 3u}), "s"(base));
  }
}

static hipEvent_t ea, eb;
template <class F> static float med_us(F f) {
  std::vector<float> v; f(); f();
  for (int r = 0; r < 7; ++r) {
    hipEventRecord(ea, 0); f(); hipEventRecord(eb, 0); hipEventSynchronize(eb);
    float ms; hipEventElapsedTime(&ms, ea, eb); v.push_back(ms * 1e3f);
  }
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char** argv) {
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  struct Shape { const char* name; uint64_t worlds, world_bytes; uint32_t span; };
  const Shape shapes[] = {{"clean_up WORLD.RGB x 4096 (120,960 B a world, spans of 11,520 B)", 4096, 120960, 11520},
                          {"clean_up per-agent RGB x 4096 (7 x 23,232 B a world, spans of 10,560 B)", 4096, 7 * 23232, 10560},
                          {"commons_harvest per-agent RGB x 4096 (16 x 23,232 B a world)", 4096, 16 * 23232, 10560}};
  const Shape sh = shapes[argc > 1 ? atoi(argv[1]) : 0];
  const uint64_t bytes = sh.worlds * sh.world_bytes;
  std::vector<uint8_t*> bufs; std::vector<std::string> names;
  for (int b = 0; b < 2; ++b) { uint8_t* p; CK(hipMalloc((void**)&p, bytes)); bufs.push_back(p); names.push_back("malloc" + std::to_string(b)); }
  { uint8_t* p = nullptr;
    if (hipExtMallocWithFlags((void**)&p, bytes, hipDeviceMallocContiguous) == hipSuccess && p) { bufs.push_back(p); names.push_back("contig"); }
    else (void)hipGetLastError(); }
  { hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    const size_t chunk = 2u << 20, n = (bytes + chunk - 1) / chunk;
    for (int v = 0; v < 3; ++v) {
      void* va = nullptr; CK(hipMemAddressReserve(&va, n * chunk, chunk, nullptr, 0));
      for (size_t i = 0; i < n; ++i) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); CK(hipMemMap((uint8_t*)va + i * chunk, chunk, 0, h, 0)); }
      hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
      CK(hipMemSetAccess(va, n * chunk, &acc, 1));
      bufs.push_back((uint8_t*)va); names.push_back("vmm2_" + std::to_string(v));
    } }
  printf("## %s\n\nus per launch (median of 7), nt stores; p = the product's order (every workgroup a contiguous range of whole worlds), f = one chip-wide front of 4 KiB spans\n\n", sh.name);
  printf("| buffer | memset |");
  const int Gs[] = {128, 192, 228, 256}, Ws[] = {4, 6, 8, 10, 13};
  for (int G : Gs) for (int W : Ws) printf(" %dx%d p | f |", G, W);
  printf("\n|---|---:|"); for (int i = 0; i < 20; ++i) printf("---:|---:|"); printf("\n");
  for (size_t b = 0; b < bufs.size(); ++b) {
    uint8_t* p = bufs[b];
    printf("| %s | %.1f |", names[b].c_str(), med_us([&] { (void)hipMemsetAsync(p, 1, bytes, 0); }));
    for (int G : Gs) for (int W : Ws) {
      const uint64_t own = (sh.worlds + G - 1) / G * sh.world_bytes;
      const float a = med_us([&] { hipLaunchKernelGGL(k_fill, dim3(G), dim3(W * 64), 0, 0, p, bytes, own, sh.span, 0); });
      const float f = med_us([&] { hipLaunchKernelGGL(k_fill, dim3(G), dim3(W * 64), 0, 0, p, bytes, (uint64_t)0, 4096u, 1); });
      printf(" %.1f | %.1f |", a, f);
    }
    printf("\n"); fflush(stdout);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
