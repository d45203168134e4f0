// Which ORDER of spans makes a persistent writer insensitive to where its buffer lies?
//
//   hipcc --offload-arch=gfx950 -O3 -o write_fronts write_fronts.hip && ./write_fronts [shape]
//
// profiles/r03_buffer_placement.md: the frame kernel — one persistent workgroup per CU,
// each streaming through its own contiguous multi-MB range of the output, span by span
// (a span = one pass of a renderer wave, 10 - 12 KB) — runs 99 - 122 us (clean_up) or
// 269 - 355 us (commons_harvest) depending on the buffer, a grid-stride fill writes
// every buffer at 6.2 - 6.9 TB/s.  This benchmark keeps the frame kernel's store form
// (8 - 10 storing waves per workgroup, 16-byte lane-contiguous chunks, whole spans per
// wave, tickets from an LDS counter) and varies only the ORDER in which the chip's
// workgroups visit the spans — a host-built table per workgroup — over several
// ordinary buffers and one physically contiguous extent (the deterministic worst
// case of round 3).  A schedule is worth building into k_frame if its slowest buffer
// is as fast as the product order's fastest.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kNone = 0xffffffffu;
constexpr int kMaxOrder = 12 * 1024;   // table entries per workgroup (48 KB of LDS)

template <int kPolicy>   // 0 plain, 1 nt, 2 sc1
__device__ inline void store16(uint8_t* base, uint32_t off, u32x4 v) {
  if (kPolicy == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(off), "v"(v), "s"(base));
  else if (kPolicy == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" :: "v"(off), "v"(v), "s"(base));
  else asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(off), "v"(v), "s"(base));
}

// stamps: per workgroup {start, first wave past 1/4, 1/2, 3/4 of the tickets, last wave out}
// (wall_clock64, 100 MHz); busy: dependent multiply-adds per span before its stores (the
// renderer's phase 1 + LDS reads: a wave of k_frame stores 12 KiB every 3 - 4.5 us, not
// back to back)
template <int kPolicy>
__global__ __launch_bounds__(1024) void k_sched(uint8_t* out, const uint32_t* order, uint32_t per,
                                                uint32_t span_bytes, unsigned long long* stamps,
                                                uint32_t busy) {
  __shared__ uint32_t next, left;
  if (threadIdx.x == 0) { left = blockDim.x >> 6; if (stamps) stamps[blockIdx.x * 5] = wall_clock64(); }
  __shared__ uint32_t tab[kMaxOrder];
  for (uint32_t i = threadIdx.x; i < per; i += blockDim.x) tab[i] = order[(size_t)blockIdx.x * per + i];
  if (threadIdx.x == 0) next = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int n_iters = (int)((span_bytes + 1023u) >> 10);
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&next, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (t >= per) break;
    if (stamps && lane == 0 && t > 0 && (t * 4) % per < 4 && t * 4 / per >= 1 && t * 4 / per <= 3)
      stamps[blockIdx.x * 5 + t * 4 / per] = wall_clock64();
    const uint32_t s_idx = tab[t];
    if (s_idx == kNone) continue;
    uint32_t spin = t;
    for (uint32_t i = 0; i < busy; ++i) spin = spin * 1664525u + 1013904223u;
    asm volatile("" :: "v"(spin));
    uint8_t* span = out + (size_t)s_idx * span_bytes;
    const uint64_t sp = reinterpret_cast<uint64_t>(span);
    span = reinterpret_cast<uint8_t*>(
        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sp));
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      if (it >= n_iters) break;
      const uint32_t off = (uint32_t)(it * 64 + lane) * 16u;
      if (off < span_bytes) store16<kPolicy>(span, off, u32x4{t, off, 2u, 3u});
    }
  }
  if (stamps && lane == 0 && atomicSub(&left, 1u) == 1u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamps[blockIdx.x * 5 + 4] = wall_clock64();
  }
}

// Dynamic balance: the buffer is cut into chunks of `c` spans; workgroup g owns chunks
// [g * ks, g * ks + ks) (a contiguous range) and, once through them, claims further
// chunks one at a time from a device-wide counter (chunks G * ks ... in ascending order).
// The counter is monotonic over launches (`base` = its value when this launch starts):
// nothing to reset.  The wave that draws the first ticket of chunk k claims chunk k + 1
// (one chunk ahead: the atomic's latency is off the critical path).
template <int kPolicy>
__global__ __launch_bounds__(1024) void k_dyn(uint8_t* out, unsigned int* counter, unsigned int base,
                                              uint32_t nchunks, uint32_t c, uint32_t ks,
                                              uint32_t span_bytes, uint32_t nspans,
                                              unsigned long long* stamps) {
  __shared__ uint32_t next, left;
  __shared__ uint32_t cb[256];   // chunk id of this workgroup's k-th chunk, + 1 (0 = not yet known)
  constexpr uint32_t kEnd = 0xffffffffu;
  const uint32_t G = gridDim.x, g = blockIdx.x;
  auto claim = [&](uint32_t k) -> uint32_t {   // one lane
    if (k < ks) return g * ks + k + 1u;
    const uint32_t id = G * ks + (atomicAdd(counter, 1u) - base);
    return id < nchunks ? id + 1u : kEnd;
  };
  if (threadIdx.x < 256) cb[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    next = 0; left = blockDim.x >> 6;
    if (stamps) stamps[g * 5] = wall_clock64();
    cb[0] = claim(0);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int n_iters = (int)((span_bytes + 1023u) >> 10);
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&next, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    const uint32_t k = t / c, i = t - k * c;
    if (k >= 255u) break;
    if (i == 0 && lane == 0) {
      const uint32_t prev = __hip_atomic_load(&cb[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)prev;
      // (cb[k] is known here for k == 0; for k > 0 it was claimed by the first ticket of k - 1,
      // which was drawn earlier — wait for it below like everybody else)
    }
    uint32_t id1;
    for (;;) {
      id1 = __hip_atomic_load(&cb[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (id1 != 0) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (i == 0 && lane == 0) {
      const uint32_t nx = id1 == kEnd ? kEnd : claim(k + 1);
      __hip_atomic_store(&cb[k + 1], nx, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (id1 == kEnd) break;
    const uint32_t s_idx = (id1 - 1u) * c + i;
    if (s_idx >= nspans) continue;
    uint8_t* span = out + (size_t)s_idx * span_bytes;
    const uint64_t sp = reinterpret_cast<uint64_t>(span);
    span = reinterpret_cast<uint8_t*>(
        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sp));
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      if (it >= n_iters) break;
      const uint32_t off = (uint32_t)(it * 64 + lane) * 16u;
      if (off < span_bytes) store16<kPolicy>(span, off, u32x4{t, off, 2u, 3u});
    }
  }
  if (stamps && lane == 0 && atomicSub(&left, 1u) == 1u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamps[g * 5 + 4] = wall_clock64();
  }
}

// Range stealing: workgroup g owns the contiguous range of chunks [g * pr, (g + 1) * pr)
// and claims them front to back from next[g]; once its range is empty it claims from
// the range with the most chunks left (anybody's front: owner and thieves share one
// counter per range).  Contiguous while a workgroup is on its own range, balanced at
// the end.  `next` is zeroed by the host before every launch here (a product kernel
// would alternate between two sets of counters, each launch zeroing the other's).
template <int kPolicy>
__global__ __launch_bounds__(1024) void k_steal(uint8_t* out, unsigned int* nextw, uint32_t nchunks,
                                                uint32_t c, uint32_t pr, uint32_t span_bytes,
                                                uint32_t nspans, unsigned long long* stamps) {
  __shared__ uint32_t next, left, stolen;
  __shared__ uint32_t cb[256];
  constexpr uint32_t kEnd = 0xffffffffu;
  const uint32_t G = gridDim.x, g = blockIdx.x;
  const int lane = threadIdx.x & 63;
  // called by one whole wave (uniform): the claim itself by lane 0, the victim search by all
  auto claim_wave = [&]() -> uint32_t {
    uint32_t got = kEnd;
    {
      uint32_t id = 0;
      if (lane == 0) id = atomicAdd(&nextw[g], 1u);
      id = (uint32_t)__builtin_amdgcn_readfirstlane((int)id);
      const uint32_t chunk = g * pr + id;
      if (id < pr && chunk < nchunks) return chunk + 1u;
    }
    for (int attempt = 0; attempt < 4 && got == kEnd; ++attempt) {
      // remaining chunks of every range (4 ranges per lane), most first
      uint32_t best_rem = 0, best_r = 0;
      for (uint32_t r = lane; r < G; r += 64) {
        const uint32_t taken = __hip_atomic_load(&nextw[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t have = pr;
        if ((uint64_t)r * pr + have > nchunks) have = nchunks > r * pr ? nchunks - r * pr : 0;
        const uint32_t rem = taken < have ? have - taken : 0;
        if (rem > best_rem) { best_rem = rem; best_r = r; }
      }
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t orem = __shfl_xor(best_rem, o), orr = __shfl_xor(best_r, o);
        if (orem > best_rem || (orem == best_rem && orr < best_r)) { best_rem = orem; best_r = orr; }
      }
      if (best_rem == 0) break;
      uint32_t id = 0;
      if (lane == 0) id = atomicAdd(&nextw[best_r], 1u);
      id = (uint32_t)__builtin_amdgcn_readfirstlane((int)id);
      uint32_t have = pr;
      if ((uint64_t)best_r * pr + have > nchunks) have = nchunks - best_r * pr;
      if (id < have) { got = best_r * pr + id + 1u; if (lane == 0) atomicAdd(&stolen, 1u); }
    }
    return got;
  };
  if (threadIdx.x < 256) cb[threadIdx.x] = 0;
  if (threadIdx.x == 0) { next = 0; stolen = 0; left = blockDim.x >> 6; if (stamps) stamps[g * 5] = wall_clock64(); }
  __syncthreads();
  if (threadIdx.x < 64) { const uint32_t v = claim_wave(); if (lane == 0) cb[0] = v; }
  __syncthreads();
  const int n_iters = (int)((span_bytes + 1023u) >> 10);
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&next, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    const uint32_t k = t / c, i = t - k * c;
    if (k >= 255u) break;
    uint32_t id1;
    for (;;) {
      id1 = __hip_atomic_load(&cb[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (id1 != 0) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (i == 0) {   // this wave drew the first ticket of chunk k: it claims chunk k + 1
      const uint32_t nx = id1 == kEnd ? kEnd : claim_wave();
      if (lane == 0) __hip_atomic_store(&cb[k + 1], nx, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (id1 == kEnd) break;
    const uint32_t s_idx = (id1 - 1u) * c + i;
    if (s_idx >= nspans) continue;
    uint8_t* span = out + (size_t)s_idx * span_bytes;
    const uint64_t sp = reinterpret_cast<uint64_t>(span);
    span = reinterpret_cast<uint8_t*>(
        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sp));
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      if (it >= n_iters) break;
      const uint32_t off = (uint32_t)(it * 64 + lane) * 16u;
      if (off < span_bytes) store16<kPolicy>(span, off, u32x4{t, off, 2u, 3u});
    }
  }
  if (stamps && lane == 0 && atomicSub(&left, 1u) == 1u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamps[g * 5 + 4] = wall_clock64();
    stamps[g * 5 + 1] = stolen;
  }
}

__global__ void k_stride(uint4* out, uint64_t nvec) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nvec;
       i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

static hipEvent_t ea, eb;
template <class F>
static void time_it(F f, int reps, float* best, float* med) {
  std::vector<float> v;
  f(); f();
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(ea, 0); f(); hipEventRecord(eb, 0); hipEventSynchronize(eb);
    float ms; hipEventElapsedTime(&ms, ea, eb);
    v.push_back(ms * 1e3f);
  }
  std::sort(v.begin(), v.end());
  *best = v[0]; *med = v[v.size() / 2];
}

struct Shape { const char* name; uint64_t bytes; uint32_t span; int groups; int waves; };
// A schedule fills order[g * per + i] (kNone = nothing); returns per.
typedef std::function<uint32_t(int G, uint32_t nspans, int waves, std::vector<uint32_t>& order)> Sched;

static uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// every workgroup a contiguous range; `place(g)` = which range workgroup g owns
static Sched contiguous(std::function<int(int, int)> place, int rotate = 0, int shuffle_chunk = 0, bool alt_dir = false) {
  return [=](int G, uint32_t nspans, int, std::vector<uint32_t>& order) {
    const uint32_t per = (nspans + G - 1) / G;
    order.assign((size_t)G * per, kNone);
    for (int g = 0; g < G; ++g) {
      const uint32_t base = (uint32_t)place(g, G) * per;
      std::vector<uint32_t> idx(per);
      for (uint32_t i = 0; i < per; ++i) idx[i] = i;
      if (rotate) { const uint32_t rot = mix(g * 2654435761u + 17u) % per; for (uint32_t i = 0; i < per; ++i) idx[i] = (i + rot) % per; }
      if (alt_dir && (g & 1)) std::reverse(idx.begin(), idx.end());
      if (shuffle_chunk) {   // Fisher-Yates over chunks of shuffle_chunk spans
        const uint32_t nc = (per + shuffle_chunk - 1) / shuffle_chunk;
        std::vector<uint32_t> cs(nc);
        for (uint32_t c = 0; c < nc; ++c) cs[c] = c;
        uint32_t seed = mix(g + 99u);
        for (uint32_t c = nc - 1; c > 0; --c) { seed = mix(seed + c); std::swap(cs[c], cs[seed % (c + 1)]); }
        std::vector<uint32_t> out2;
        for (uint32_t c = 0; c < nc; ++c)
          for (uint32_t i = cs[c] * shuffle_chunk; i < std::min<uint32_t>(per, (cs[c] + 1) * shuffle_chunk); ++i) out2.push_back(i);
        idx = out2;
      }
      for (uint32_t i = 0; i < per; ++i) {
        const uint32_t s = base + idx[i];
        order[(size_t)g * per + i] = s < nspans ? s : kNone;
      }
    }
    return per;
  };
}

// teams of T workgroups share a contiguous range and walk it together, `cw` consecutive
// spans per member and round (cw = 0: one per storing wave).  by_xcd: a team's members
// sit on one XCD (workgroup g runs on XCD g % 8; T >= the XCD's workgroups = the whole
// XCD is one team); otherwise members are T consecutive workgroups.
static Sched teams(int T, int cw_or_0, bool by_xcd, int cw_wave_sets = 0) {
  return [=](int G, uint32_t nspans, int waves, std::vector<uint32_t>& order) {
    const int cw = cw_wave_sets ? cw_wave_sets * waves : cw_or_0 ? cw_or_0 : waves;
    std::vector<std::vector<int>> tm;
    if (by_xcd) {
      for (int x = 0; x < 8; ++x) {
        std::vector<int> mem;
        for (int g = x; g < G; g += 8) mem.push_back(g);
        for (size_t i = 0; i < mem.size(); i += T)
          tm.emplace_back(mem.begin() + i, mem.begin() + std::min(mem.size(), i + (size_t)T));
      }
    } else {
      for (int g = 0; g < G; g += T) {
        std::vector<int> mem;
        for (int h = g; h < std::min(G, g + T); ++h) mem.push_back(h);
        tm.push_back(mem);
      }
    }
    const uint32_t per0 = (nspans + G - 1) / G;
    const uint32_t per = (per0 + cw - 1) / cw * cw;
    order.assign((size_t)G * per, kNone);
    uint32_t start = 0;
    for (auto& mem : tm) {
      const uint32_t TT = (uint32_t)mem.size(), tper = TT * per0;
      for (uint32_t j = 0; j < TT; ++j)
        for (uint32_t i = 0; i < per; ++i) {
          const uint32_t r = i / cw, k = i % cw;
          const uint32_t local = (r * TT + j) * cw + k;
          const uint32_t s = start + local;
          order[(size_t)mem[j] * per + i] = (local < tper && s < nspans) ? s : kNone;
        }
      start += tper;
    }
    return per;
  };
}

int main(int argc, char** argv) {
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  const Shape shapes[] = {
      {"clean_up WORLD.RGB 4096 x 120,960 B, spans of 11,520 B, 256 x 8 waves", 4096ull * 120960, 11520, 256, 8},
      {"commons_harvest 4096 x 16 x 23,232 B, spans of 10,560 B, 228 x 10 waves", 4096ull * 16 * 23232, 10560, 228, 10},
      {"clean_up bytes in 1 KiB spans (a persistent grid-stride fill), 256 x 8 waves", 4096ull * 120960, 1024, 256, 8},
      {"clean_up bytes in 4 KiB spans, 256 x 8 waves", 4096ull * 120960, 4096, 256, 8},
  };
  const int which = argc > 1 ? atoi(argv[1]) : 0;
  const Shape sh = shapes[which];
  const int policy = argc > 2 ? atoi(argv[2]) : 1;
  const int nplain = getenv("NBUF") ? atoi(getenv("NBUF")) : 6;
  std::vector<uint8_t*> bufs; std::vector<std::string> bname;
  for (int b = 0; b < nplain; ++b) {
    uint8_t* p; CK(hipMalloc((void**)&p, sh.bytes)); bufs.push_back(p); bname.push_back("malloc" + std::to_string(b));
  }
  if (!getenv("NO_CONTIG")) {
    uint8_t* p = nullptr;
    if (hipExtMallocWithFlags((void**)&p, sh.bytes, hipDeviceMallocContiguous) == hipSuccess && p) { bufs.push_back(p); bname.push_back("contig"); }
    else { (void)hipGetLastError(); printf("(no physically contiguous extent)\n"); }
  }
  // VMM=<chunk MB>[,<chunk MB>...]: per chunk size, ONE set of physical chunks mapped
  // three times — in creation order, reversed, shuffled: same memory, another layout
  if (const char* vmm = getenv("VMM")) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    for (const char* q = vmm; *q;) {
      // (MB; a trailing k = KB)
      size_t chunk = (size_t)atoi(q) << 20;
      { const char* z = q; while (*z && *z != ',') ++z; if (z > q && z[-1] == 'k') chunk >>= 10; }
      while (*q && *q != ',') ++q; if (*q == ',') ++q;
      if (chunk == 0 || chunk % gran) { printf("(chunk of %zu B: granularity is %zu)\n", chunk, gran); continue; }
      const size_t n = (sh.bytes + chunk - 1) / chunk;
      std::vector<hipMemGenericAllocationHandle_t> hs(n);
      bool ok = true;
      for (size_t i = 0; i < n && ok; ++i) ok = hipMemCreate(&hs[i], chunk, &prop, 0) == hipSuccess;
      if (!ok) { (void)hipGetLastError(); printf("(hipMemCreate failed for %zu MB chunks)\n", chunk >> 20); continue; }
      printf("(granularity %zu B)\n", gran);
      for (int layout = 0; layout < (getenv("VMM_LAYOUTS") ? atoi(getenv("VMM_LAYOUTS")) : 3); ++layout) {
        std::vector<size_t> perm(n);
        for (size_t i = 0; i < n; ++i) perm[i] = layout == 1 ? n - 1 - i : i;
        if (layout == 2) { uint32_t seed = 12345; for (size_t i = n - 1; i > 0; --i) { seed = mix(seed + (uint32_t)i); std::swap(perm[i], perm[seed % (i + 1)]); } }
        void* va = nullptr; CK(hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0));
        for (size_t i = 0; i < n; ++i) CK(hipMemMap((uint8_t*)va + i * chunk, chunk, 0, hs[perm[i]], 0));
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, n * chunk, &acc, 1));
        bufs.push_back((uint8_t*)va);
        bname.push_back("vmm" + (chunk >= (1u << 20) ? std::to_string(chunk >> 20) : std::to_string(chunk >> 10) + "k") + (layout == 0 ? "" : layout == 1 ? "rev" : "shuf"));
      }
    }
  }
  const uint32_t nspans = (uint32_t)(sh.bytes / sh.span);
  struct V { std::string name; Sched s; };
  auto ident = [](int g, int) { return g; };
  auto xcd_major = [](int g, int G) { return G % 8 == 0 ? (g % 8) * (G / 8) + g / 8 : g; };
  std::vector<V> vs = {
      {"contiguous per workgroup (product)", contiguous(ident)},
      {"contiguous, ranges XCD-major", contiguous(xcd_major)},
      {"contiguous, rotated start", contiguous(ident, 1)},
      {"contiguous, odd workgroups backwards", contiguous(ident, 0, 0, true)},
      {"contiguous, chunks of 8 spans shuffled", contiguous(ident, 0, 8)},
      {"contiguous, spans shuffled", contiguous(ident, 0, 1)},
      {"one front: all workgroups, 1 span each", teams(1 << 20, 1, false)},
      {"one front: all workgroups, a wave-set each", teams(1 << 20, 0, false)},
      {"one front, 2 wave-sets each", teams(1 << 20, 0, false, 2)},
      {"8 XCD teams, a wave-set each", teams(1 << 20, 0, true)},
      {"teams of 32 on one XCD", teams(32, 0, true)},
      {"teams of 16 on one XCD", teams(16, 0, true)},
      {"teams of 8 on one XCD", teams(8, 0, true)},
      {"teams of 4 on one XCD", teams(4, 0, true)},
      {"teams of 2 on one XCD", teams(2, 0, true)},
      {"teams of 32 consecutive workgroups", teams(32, 0, false)},
      {"teams of 8 consecutive workgroups", teams(8, 0, false)},
      {"teams of 32 on one XCD, 1 span each", teams(32, 1, true)},
      {"8 XCD teams, 2 wave-sets each", teams(1 << 20, 0, true, 2)},
      {"8 XCD teams, 4 wave-sets each", teams(1 << 20, 0, true, 4)},
  };
  printf("## %s, %s stores\n\n| order |", sh.name, policy == 1 ? "nt" : policy == 2 ? "sc1" : "plain");
  for (auto& n : bname) printf(" %s |", n.c_str());
  printf(" min - max TB/s |\n|---|"); for (size_t b = 0; b <= bufs.size(); ++b) printf("---:|"); printf("\n");
  {
    printf("| grid-stride fill 16384 x 1024 |");
    float lo = 1e30f, hi = 0;
    for (auto* p : bufs) {
      float best, med; time_it([&] { hipLaunchKernelGGL(k_stride, dim3(16384), dim3(1024), 0, 0, (uint4*)p, sh.bytes / 16); }, 8, &best, &med);
      printf(" %.1f |", med); lo = std::min(lo, med); hi = std::max(hi, med);
    }
    printf(" %.2f - %.2f |\n", sh.bytes / hi / 1e6, sh.bytes / lo / 1e6);
    printf("| hipMemsetAsync |");
    lo = 1e30f; hi = 0;
    for (auto* p : bufs) {
      float best, med; time_it([&] { (void)hipMemsetAsync(p, 1, sh.bytes, 0); }, 8, &best, &med);
      printf(" %.1f |", med); lo = std::min(lo, med); hi = std::max(hi, med);
    }
    printf(" %.2f - %.2f |\n", sh.bytes / hi / 1e6, sh.bytes / lo / 1e6);
  }
  uint32_t* d_order = nullptr; size_t d_cap = 0;
  const uint32_t busy = getenv("BUSY") ? (uint32_t)atoi(getenv("BUSY")) : 0u;
  if (busy) printf("(busy work per span: %u dependent multiply-adds)\n", busy);
  const char* only = getenv("VARIANTS");   // e.g. "0,6,17"
  unsigned long long* d_stamps = nullptr;
  const bool want_stamps = getenv("STAMPS") != nullptr;
  if (want_stamps) CK(hipMalloc((void**)&d_stamps, sizeof(unsigned long long) * 5 * sh.groups));
  std::vector<std::string> stamp_report;
  for (size_t vi = 0; vi < vs.size(); ++vi) {
    auto& v = vs[vi];
    if (only) {
      bool hit = false;
      for (const char* q = only; *q;) { if ((size_t)atoi(q) == vi) hit = true; while (*q && *q != ',') ++q; if (*q == ',') ++q; }
      if (!hit) continue;
    }
    std::vector<uint32_t> order;
    const uint32_t per = v.s(sh.groups, nspans, sh.waves, order);
    if (per > (uint32_t)kMaxOrder) { printf("| %s | (table too big: %u) |\n", v.name.c_str(), per); continue; }
    // every span exactly once?
    {
      std::vector<uint8_t> seen(nspans, 0); size_t n = 0; bool dup = false;
      for (uint32_t s : order) if (s != kNone) { if (s >= nspans || seen[s]) dup = true; else { seen[s] = 1; ++n; } }
      if (dup || n != nspans) { printf("| %s | (bad schedule: %zu of %u spans%s) |\n", v.name.c_str(), n, nspans, dup ? ", duplicates" : ""); continue; }
    }
    if (order.size() * 4 > d_cap) { if (d_order) CK(hipFree(d_order)); d_cap = order.size() * 4; CK(hipMalloc((void**)&d_order, d_cap)); }
    CK(hipMemcpy(d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice));
    printf("| %s |", v.name.c_str());
    float lo = 1e30f, hi = 0;
    for (size_t bi = 0; bi < bufs.size(); ++bi) {
      uint8_t* p = bufs[bi];
      float best, med;
      unsigned long long* st = nullptr;
      auto launch = [&] {
        if (policy == 1) hipLaunchKernelGGL(k_sched<1>, dim3(sh.groups), dim3(sh.waves * 64), 0, 0, p, d_order, per, sh.span, st, busy);
        else if (policy == 2) hipLaunchKernelGGL(k_sched<2>, dim3(sh.groups), dim3(sh.waves * 64), 0, 0, p, d_order, per, sh.span, st, busy);
        else hipLaunchKernelGGL(k_sched<0>, dim3(sh.groups), dim3(sh.waves * 64), 0, 0, p, d_order, per, sh.span, st, busy);
      };
      time_it(launch, 8, &best, &med);
      printf(" %.1f |", med); lo = std::min(lo, med); hi = std::max(hi, med);
      if (want_stamps) {
        // one more launch with stamps: when does each workgroup pass 1/4, 1/2, 3/4, end?
        st = d_stamps; CK(hipMemset(d_stamps, 0, sizeof(unsigned long long) * 5 * sh.groups));
        launch(); CK(hipDeviceSynchronize()); st = nullptr;
        std::vector<unsigned long long> hsb(5 * sh.groups);
        CK(hipMemcpy(hsb.data(), d_stamps, hsb.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; for (int g = 0; g < sh.groups; ++g) t0 = std::min(t0, hsb[g * 5]);
        char line[1024]; int n = snprintf(line, sizeof line, "| %s | %s |", v.name.c_str(), bname[bi].c_str());
        for (int q = 1; q <= 4; ++q) {
          std::vector<double> e;
          for (int g = 0; g < sh.groups; ++g) if (hsb[g * 5 + q]) e.push_back((double)(hsb[g * 5 + q] - t0) * 0.01);
          std::sort(e.begin(), e.end());
          if (e.empty()) { n += snprintf(line + n, sizeof line - n, " - |"); continue; }
          n += snprintf(line + n, sizeof line - n, " %.1f / %.1f / %.1f / %.1f |", e[0], e[e.size() / 2], e[e.size() * 9 / 10], e.back());
        }
        // end time by XCD (mean), and the start skew
        double sx[8] = {}, nx[8] = {}; double smax = 0;
        for (int g = 0; g < sh.groups; ++g) { sx[g % 8] += (double)(hsb[g * 5 + 4] - t0) * 0.01; nx[g % 8] += 1; smax = std::max(smax, (double)(hsb[g * 5] - t0) * 0.01); }
        for (int x = 0; x < 8; ++x) n += snprintf(line + n, sizeof line - n, " %.0f", nx[x] ? sx[x] / nx[x] : 0.0);
        n += snprintf(line + n, sizeof line - n, " | %.1f |", smax);
        stamp_report.push_back(line);
      }
    }
    printf(" %.2f - %.2f |\n", sh.bytes / hi / 1e6, sh.bytes / lo / 1e6);
    fflush(stdout);
  }
  // ---- calibrated static split: contiguous ranges whose LENGTHS follow the rates the
  // XCDs were seen to write this buffer at (per-workgroup end stamps of the previous
  // launch); three rounds of feedback per buffer
  if (getenv("CALIBRATE")) {
    const bool per_wg = atoi(getenv("CALIBRATE")) == 2;
    printf("\ncalibrated split (%s shares), us per launch and [first .. last workgroup end]: even split, then rounds 1 - 4\n\n| buffer | even | 1 | 2 | 3 | 4 |\n|---|---|---|---|---|---|\n", per_wg ? "per-workgroup" : "per-XCD");
    if (!d_stamps) CK(hipMalloc((void**)&d_stamps, sizeof(unsigned long long) * 5 * sh.groups));
    for (size_t bi = 0; bi < bufs.size(); ++bi) {
      uint8_t* p = bufs[bi];
      const int G = sh.groups;
      std::vector<double> share(G, (double)nspans / G);
      printf("| %s |", bname[bi].c_str());
      for (int round = 0; round < 5; ++round) {
        // ranges from the shares (whole spans), in workgroup order
        std::vector<uint32_t> start(G + 1, 0);
        double acc = 0;
        for (int g = 0; g < G; ++g) { acc += share[g]; start[g + 1] = (uint32_t)std::min<double>(nspans, acc + 0.5); }
        start[G] = nspans;
        uint32_t per = 0;
        for (int g = 0; g < G; ++g) per = std::max(per, start[g + 1] - start[g]);
        if (per > (uint32_t)kMaxOrder) { printf(" (table too big) |"); break; }
        std::vector<uint32_t> order((size_t)G * per, kNone);
        for (int g = 0; g < G; ++g)
          for (uint32_t i = 0; i < start[g + 1] - start[g]; ++i) order[(size_t)g * per + i] = start[g] + i;
        if (order.size() * 4 > d_cap) { if (d_order) CK(hipFree(d_order)); d_cap = order.size() * 4; CK(hipMalloc((void**)&d_order, d_cap)); }
        CK(hipMemcpy(d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice));
        unsigned long long* st = nullptr;
        auto launch = [&] {
          if (policy == 1) hipLaunchKernelGGL(k_sched<1>, dim3(G), dim3(sh.waves * 64), 0, 0, p, d_order, per, sh.span, st, 0u);
          else if (policy == 2) hipLaunchKernelGGL(k_sched<2>, dim3(G), dim3(sh.waves * 64), 0, 0, p, d_order, per, sh.span, st, 0u);
          else hipLaunchKernelGGL(k_sched<0>, dim3(G), dim3(sh.waves * 64), 0, 0, p, d_order, per, sh.span, st, 0u);
        };
        float best, med; time_it(launch, 8, &best, &med);
        // stamps of three more launches, averaged
        std::vector<double> e(G, 0.0);
        for (int rep = 0; rep < 3; ++rep) {
          st = d_stamps; CK(hipMemset(d_stamps, 0, sizeof(unsigned long long) * 5 * G));
          launch(); CK(hipDeviceSynchronize()); st = nullptr;
          std::vector<unsigned long long> hsb(5 * G);
          CK(hipMemcpy(hsb.data(), d_stamps, hsb.size() * 8, hipMemcpyDeviceToHost));
          unsigned long long t0 = ~0ull; for (int g = 0; g < G; ++g) t0 = std::min(t0, hsb[g * 5]);
          for (int g = 0; g < G; ++g) e[g] += (double)(hsb[g * 5 + 4] - t0) * 0.01 / 3;
        }
        printf(" %.1f [%.0f .. %.0f] |", med, *std::min_element(e.begin(), e.end()), *std::max_element(e.begin(), e.end()));
        // new shares: rate = share / end time, per XCD (or per workgroup), renormalised
        std::vector<double> rate(G);
        if (per_wg) for (int g = 0; g < G; ++g) rate[g] = share[g] / e[g];
        else {
          double rs[8] = {}, rn[8] = {};
          for (int g = 0; g < G; ++g) { rs[g % 8] += share[g] / e[g]; rn[g % 8] += 1; }
          for (int g = 0; g < G; ++g) rate[g] = rs[g % 8] / rn[g % 8];
        }
        double tot = 0; for (double r : rate) tot += r;
        for (int g = 0; g < G; ++g) share[g] = 0.5 * share[g] + 0.5 * rate[g] / tot * nspans;   // (damped)
      }
      printf("\n"); fflush(stdout);
    }
  }
  // ---- dynamic balance
  if (!getenv("NO_DYN")) {
    unsigned int* d_counter; CK(hipMalloc((void**)&d_counter, 4)); CK(hipMemset(d_counter, 0, 4));
    unsigned int h_base = 0;
    const uint32_t per0 = (nspans + sh.groups - 1) / sh.groups;   // a workgroup's fair share
    struct D { uint32_t c; int static_pct; };
    const uint32_t c4 = (per0 + 3) / 4, c8 = (per0 + 7) / 8, c16 = (per0 + 15) / 16;
    const D ds[] = {{c4, 100}, {c4, 75}, {c4, 50}, {c4, 0}, {c8, 75}, {c8, 50}, {c8, 0}, {c16, 75}, {c16, 50}, {c16, 0}};
    for (const D& d : ds) {
      const uint32_t nchunks = (nspans + d.c - 1) / d.c;
      const uint32_t fair = (nchunks + sh.groups - 1) / sh.groups;
      uint32_t ks = fair * d.static_pct / 100;
      if (d.static_pct == 100) ks = fair;
      if ((uint64_t)ks * sh.groups > nchunks) ks = nchunks / sh.groups;
      char name[160]; snprintf(name, sizeof name, "chunks of %u spans: %u static per workgroup + the rest (%u of %u) dynamic", d.c, ks, nchunks - ks * sh.groups, nchunks);
      printf("| %s |", name);
      float lo = 1e30f, hi = 0;
      for (size_t bi = 0; bi < bufs.size(); ++bi) {
        uint8_t* p = bufs[bi];
        unsigned long long* st = nullptr;
        const uint32_t dynamic = nchunks - ks * sh.groups;
        auto launch = [&] {
          if (policy == 1) hipLaunchKernelGGL(k_dyn<1>, dim3(sh.groups), dim3(sh.waves * 64), 0, 0, p, d_counter, h_base, nchunks, d.c, ks, sh.span, nspans, st);
          else if (policy == 2) hipLaunchKernelGGL(k_dyn<2>, dim3(sh.groups), dim3(sh.waves * 64), 0, 0, p, d_counter, h_base, nchunks, d.c, ks, sh.span, nspans, st);
          else hipLaunchKernelGGL(k_dyn<0>, dim3(sh.groups), dim3(sh.waves * 64), 0, 0, p, d_counter, h_base, nchunks, d.c, ks, sh.span, nspans, st);
          h_base += dynamic + sh.groups;   // every workgroup's last claim runs past the end once
        };
        float best, med; time_it(launch, 8, &best, &med);
        printf(" %.1f |", med); lo = std::min(lo, med); hi = std::max(hi, med);
        if (want_stamps) {
          st = d_stamps; CK(hipMemset(d_stamps, 0, sizeof(unsigned long long) * 5 * sh.groups));
          launch(); CK(hipDeviceSynchronize()); st = nullptr;
          std::vector<unsigned long long> hsb(5 * sh.groups);
          CK(hipMemcpy(hsb.data(), d_stamps, hsb.size() * 8, hipMemcpyDeviceToHost));
          unsigned long long t0 = ~0ull; for (int g = 0; g < sh.groups; ++g) t0 = std::min(t0, hsb[g * 5]);
          std::vector<double> e; double sx[8] = {}, nx[8] = {};
          for (int g = 0; g < sh.groups; ++g) { const double v = (double)(hsb[g * 5 + 4] - t0) * 0.01; e.push_back(v); sx[g % 8] += v; nx[g % 8] += 1; }
          std::sort(e.begin(), e.end());
          char line[512]; int n = snprintf(line, sizeof line, "| %s | %s | - | - | - | %.1f / %.1f / %.1f / %.1f |", name, bname[bi].c_str(), e[0], e[e.size() / 2], e[e.size() * 9 / 10], e.back());
          for (int x = 0; x < 8; ++x) n += snprintf(line + n, sizeof line - n, " %.0f", nx[x] ? sx[x] / nx[x] : 0.0);
          n += snprintf(line + n, sizeof line - n, " | |");
          stamp_report.push_back(line);
        }
      }
      printf(" %.2f - %.2f |\n", sh.bytes / hi / 1e6, sh.bytes / lo / 1e6);
      fflush(stdout);
    }
    // does every launch still write every span?  (the last configuration, on buffer 0)
    {
      CK(hipDeviceSynchronize());
      unsigned int cnt = 0; CK(hipMemcpy(&cnt, d_counter, 4, hipMemcpyDeviceToHost));
      printf("\n(claim counter %u, host's base %u: %s)\n", cnt, h_base, cnt == h_base ? "in step" : "OUT OF STEP");
    }
  }
  // ---- range stealing
  if (!getenv("NO_STEAL")) {
    unsigned int* d_next; CK(hipMalloc((void**)&d_next, 4 * 256));
    const uint32_t per0 = (nspans + sh.groups - 1) / sh.groups;
    for (uint32_t div : {4u, 8u, 16u, 32u}) {
      const uint32_t c = (per0 + div - 1) / div;
      const uint32_t nchunks = (nspans + c - 1) / c;
      const uint32_t pr = (nchunks + sh.groups - 1) / sh.groups;
      char name[160]; snprintf(name, sizeof name, "own range first (%u chunks of %u spans), then steal", pr, c);
      printf("| %s |", name);
      float lo = 1e30f, hi = 0;
      for (size_t bi = 0; bi < bufs.size(); ++bi) {
        uint8_t* p = bufs[bi];
        unsigned long long* st = nullptr;
        auto launch = [&] {
          (void)hipMemsetAsync(d_next, 0, 4 * 256, 0);
          if (policy == 1) hipLaunchKernelGGL(k_steal<1>, dim3(sh.groups), dim3(sh.waves * 64), 0, 0, p, d_next, nchunks, c, pr, sh.span, nspans, st);
          else if (policy == 2) hipLaunchKernelGGL(k_steal<2>, dim3(sh.groups), dim3(sh.waves * 64), 0, 0, p, d_next, nchunks, c, pr, sh.span, nspans, st);
          else hipLaunchKernelGGL(k_steal<0>, dim3(sh.groups), dim3(sh.waves * 64), 0, 0, p, d_next, nchunks, c, pr, sh.span, nspans, st);
        };
        float best, med; time_it(launch, 8, &best, &med);
        printf(" %.1f |", med); lo = std::min(lo, med); hi = std::max(hi, med);
        if (want_stamps) {
          st = d_stamps; CK(hipMemset(d_stamps, 0, sizeof(unsigned long long) * 5 * sh.groups));
          launch(); CK(hipDeviceSynchronize()); st = nullptr;
          std::vector<unsigned long long> hsb(5 * sh.groups);
          CK(hipMemcpy(hsb.data(), d_stamps, hsb.size() * 8, hipMemcpyDeviceToHost));
          unsigned long long t0 = ~0ull; for (int g = 0; g < sh.groups; ++g) t0 = std::min(t0, hsb[g * 5]);
          std::vector<double> e; double sx[8] = {}, nx[8] = {}, stl[8] = {};
          for (int g = 0; g < sh.groups; ++g) { const double v = (double)(hsb[g * 5 + 4] - t0) * 0.01; e.push_back(v); sx[g % 8] += v; nx[g % 8] += 1; stl[g % 8] += (double)hsb[g * 5 + 1]; }
          std::sort(e.begin(), e.end());
          char line[640]; int n = snprintf(line, sizeof line, "| %s | %s | - | - | chunks stolen by XCD:", name, bname[bi].c_str());
          for (int x = 0; x < 8; ++x) n += snprintf(line + n, sizeof line - n, " %.0f", stl[x]);
          n += snprintf(line + n, sizeof line - n, " | %.1f / %.1f / %.1f / %.1f |", e[0], e[e.size() / 2], e[e.size() * 9 / 10], e.back());
          for (int x = 0; x < 8; ++x) n += snprintf(line + n, sizeof line - n, " %.0f", nx[x] ? sx[x] / nx[x] : 0.0);
          n += snprintf(line + n, sizeof line - n, " | |");
          stamp_report.push_back(line);
        }
      }
      printf(" %.2f - %.2f |\n", sh.bytes / hi / 1e6, sh.bytes / lo / 1e6);
      fflush(stdout);
    }
  }
  if (want_stamps) {
    printf("\nworkgroup progress, us after the first workgroup started: min / median / 90 %% / max over workgroups\n\n"
           "| order | buffer | 1/4 of its spans | 1/2 | 3/4 | done | mean end by XCD 0..7 | last start |\n|---|---|---|---|---|---|---|---:|\n");
    for (auto& l : stamp_report) printf("%s\n", l.c_str());
  }
  CK(hipDeviceSynchronize());
  return 0;
}
