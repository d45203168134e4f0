// Store-ceiling sweep: what is the chip's WRITE ceiling for the frame kernel's
// output shape (495 MB of WORLD.RGB per launch, written once)?
//
//   hipcc --offload-arch=gfx950 -O3 -o store_ceiling store_ceiling.hip && ./store_ceiling
//
// Every variant writes the same 4096 x 120,960 B.  Swept: the store form (16-byte
// lane-contiguous chunks = 1 KiB per wave instruction, the renderer's copy phase;
// 12 + 12-byte row halves, its direct path; 4-byte dword stores as a floor),
// the cache-policy bits, waves per workgroup, workgroups per CU (persistent: one
// or two per CU, each walking a contiguous range in 11,520-byte spans — the
// frame kernel's pass — handed out in order; or "flat": one span per wave, as
// many workgroups as it takes), and hipMemsetAsync as the runtime's own fill.
// Output: a markdown table (profiles/r03_store_ceiling.md).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr uint32_t kSpan = 11520;                 // one pass: 2 strips x 8 rows x 720 B
constexpr uint32_t kWorld = 120960;               // WORLD.RGB of one clean_up world
constexpr uint32_t kWorlds = 4096;
constexpr uint64_t kBytes = (uint64_t)kWorld * kWorlds;
constexpr uint32_t kSpans = (uint32_t)(kBytes / kSpan);   // 43,008

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

template <int kPolicy>
__device__ inline void st16(uint8_t* base, uint32_t off, u32x4 v) {
  if (kPolicy == 0) asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(off), "v"(v), "s"(base));
  if (kPolicy == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(off), "v"(v), "s"(base));
  if (kPolicy == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" :: "v"(off), "v"(v), "s"(base));
  if (kPolicy == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" :: "v"(off), "v"(v), "s"(base));
  if (kPolicy == 4) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt" :: "v"(off), "v"(v), "s"(base));
}

// form 0: 16-byte lane-contiguous chunks; form 1: lane = cell, 8 rows x (12 + 12) B;
// form 2: dword stores, lane-contiguous
template <int kForm, int kPolicy>
__device__ inline void write_span(uint8_t* span, int lane, uint32_t tag) {
  if (kForm == 0) {
#pragma unroll
    for (int it = 0; it < 12; ++it) {
      const uint32_t off = (uint32_t)(it * 64 + lane) * 16u;
      if (off < kSpan) st16<kPolicy>(span, off, u32x4{tag, off, 2u, 3u});
    }
  } else if (kForm == 1) {
    const int sr = lane / 30, cx = lane - sr * 30;
    if (sr < 2) {
#pragma unroll
      for (int py = 0; py < 8; ++py) {
        const uint32_t off = (uint32_t)sr * 5760u + (uint32_t)py * 720u + (uint32_t)cx * 24u;
        const u32x3 a = {tag, off, 2u}, b = {3u, 4u, 5u};
        asm volatile("global_store_dwordx3 %0, %1, %3\n\tglobal_store_dwordx3 %0, %2, %3 offset:12"
                     :: "v"(off), "v"(a), "v"(b), "s"(span) : "memory");
      }
    }
  } else {
    for (uint32_t off = (uint32_t)lane * 4u; off < kSpan; off += 256u)
      *reinterpret_cast<uint32_t*>(span + off) = tag;
  }
}

// persistent: `groups` workgroups, each owns a contiguous range of spans and
// hands them to its waves in order from an LDS counter (the frame kernel's
// ticket scheme)
// kStagger: 0 = all waves equal; 1 = waves that share a SIMD get different wave
// priorities (s_setprio (wave / 4) & 3); 2 = they start 0 / 2 / 4 / 6 us apart;
// 3 = every wave raises its priority for the duration of a span's stores
template <int kForm, int kPolicy, int kStagger = 0>
__global__ void k_persistent(uint8_t* out, uint32_t spans_per_group) {
  __shared__ uint32_t next;
  if (threadIdx.x == 0) next = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (kStagger == 1) {
    switch ((wv >> 2) & 3) {
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      case 3: __builtin_amdgcn_s_setprio(3); break;
      default: break;
    }
  }
  if (kStagger == 2)
    for (int i = 0; i < (wv >> 2) * 40; ++i) __builtin_amdgcn_s_sleep(2);   // ~128 cycles each
  const uint32_t first = blockIdx.x * spans_per_group;
  uint32_t n = kSpans > first ? kSpans - first : 0;
  if (n > spans_per_group) n = spans_per_group;
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&next, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (t >= n) break;
    uint8_t* span = out + (uint64_t)(first + t) * kSpan;
    const uint64_t sp = reinterpret_cast<uint64_t>(span);
    span = reinterpret_cast<uint8_t*>(
        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sp));
    if (kStagger == 3) __builtin_amdgcn_s_setprio(2);
    write_span<kForm, kPolicy>(span, lane, t);
    if (kStagger == 3) __builtin_amdgcn_s_setprio(0);
  }
}

// split roles: a workgroup of `blockDim.x / 64` waves of which only the first
// kStorers store (taking spans from the counter); the others stand for waves that
// prepare the spans: kBusy = 0 they sleep, 1 they hammer the LDS with reads
template <int kStorers, int kBusy>
__global__ void k_split(uint8_t* out, uint32_t spans_per_group) {
  __shared__ uint32_t next;
  __shared__ uint32_t done;
  __shared__ uint32_t junk[4096];
  if (threadIdx.x == 0) { next = 0; done = 0; }
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) junk[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const uint32_t first = blockIdx.x * spans_per_group;
  uint32_t n = kSpans > first ? kSpans - first : 0;
  if (n > spans_per_group) n = spans_per_group;
  if (wv >= kStorers) {
    uint32_t acc = 0;
    while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)kStorers) {
      if (kBusy) {
        for (int i = 0; i < 64; ++i) acc += junk[(lane * 17 + i * 64 + acc) & 4095];
      } else {
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if (acc == 0xdeadbeef) out[0] = 1;
    return;
  }
  for (;;) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&next, 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    if (t >= n) break;
    uint8_t* span = out + (uint64_t)(first + t) * kSpan;
    const uint64_t sp = reinterpret_cast<uint64_t>(span);
    span = reinterpret_cast<uint8_t*>(
        ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32) |
        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sp));
    write_span<0, 0>(span, lane, t);
  }
  if (lane == 0) atomicAdd(&done, 1u);
}

// persistent, but the workgroups' ranges are INTERLEAVED: workgroup g owns chunks
// g, g + G, g + 2G ... of `chunk` spans each (chunk = 42: the four worlds of one
// frame-kernel batch; 1: span by span), so that at any time the chip writes one
// compact window instead of 256 ranges 1.9 MB apart
__global__ void k_interleaved(uint8_t* out, uint32_t chunk, uint32_t chunks_per_group) {
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
    write_span<0, 0>(span, lane, t);
  }
}

// flat: one span per wave, 4 waves per workgroup
template <int kForm, int kPolicy>
__global__ void k_flat(uint8_t* out) {
  const int lane = threadIdx.x & 63;
  const uint32_t s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= kSpans) return;
  uint8_t* span = out + (uint64_t)s * kSpan;
  const uint64_t sp = reinterpret_cast<uint64_t>(span);
  span = reinterpret_cast<uint8_t*>(
      ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sp >> 32)) << 32) |
      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sp));
  write_span<kForm, kPolicy>(span, lane, s);
}

// plain grid-stride uint4 fill, every wave sweeping the whole buffer (the
// classic "copy kernel" shape: neighbouring waves write neighbouring KiBs)
__global__ void k_stride(uint4* out, uint64_t nvec) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nvec;
       i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

static hipEvent_t ea, eb;
template <class F>
static float best_us(F launch, int reps = 8) {
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    (void)hipEventRecord(ea);
    launch();
    (void)hipEventRecord(eb);
    (void)hipEventSynchronize(eb);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ea, eb);
    if (ms * 1e3f < best) best = ms * 1e3f;
  }
  return best;
}
static void row(const char* what, float us) {
  printf("| %s | %.1f | %.2f | %.3f |\n", what, us, kBytes / (us * 1e-6) / 1e12, kBytes / (us * 1e-6) / 8e12);
  fflush(stdout);
}

template <int kForm, int kPolicy, int kStagger = 0>
static void persistent_rows(uint8_t* buf, const char* form, const char* policy, int cus) {
  char name[160];
  for (int per_cu : {1, 2})
    for (int waves : {4, 8, 12, 16}) {
      if (per_cu * waves > 32) continue;
      if (kStagger && (per_cu != 1 || waves == 4)) continue;
      const int groups = cus * per_cu;
      const uint32_t spg = (kSpans + groups - 1) / groups;
      const float us = best_us([&] { k_persistent<kForm, kPolicy, kStagger><<<groups, waves * 64>>>(buf, spg); });
      snprintf(name, sizeof name, "%s, %s, persistent %d/CU x %d waves%s", form, policy, per_cu, waves,
               kStagger == 1 ? ", priority by wave / 4" : kStagger == 2 ? ", staggered start"
               : kStagger == 3 ? ", priority raised per span" : "");
      row(name, us);
    }
}

int main() {
  uint8_t* buf;
  CK(hipMalloc((void**)&buf, kBytes));
  CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  printf("| variant (495.5 MB written once, best of 8) | us | TB/s | of 8 TB/s |\n|---|---:|---:|---:|\n");
  row("hipMemsetAsync", best_us([&] { (void)hipMemsetAsync(buf, 1, kBytes, 0); }));
  for (int g : {1024, 4096, 16384})
    for (int th : {256, 1024}) {
      char name[96];
      snprintf(name, sizeof name, "grid-stride uint4 fill, %d x %d", g, th);
      row(name, best_us([&] { k_stride<<<g, th>>>((uint4*)buf, kBytes / 16); }));
    }
  row("16 B chunks, default, flat (1 span per wave, 256-thread WGs)",
      best_us([&] { k_flat<0, 0><<<(kSpans + 3) / 4, 256>>>(buf); }));
  row("12+12 B rows, default, flat", best_us([&] { k_flat<1, 0><<<(kSpans + 3) / 4, 256>>>(buf); }));
  row("dword, default, flat", best_us([&] { k_flat<2, 0><<<(kSpans + 3) / 4, 256>>>(buf); }));
  persistent_rows<0, 0>(buf, "16 B chunks", "default", cus);
  if (getenv("INTERLEAVE")) {
    char name[160];
    for (int rep = 0; rep < 2; ++rep)
      for (int waves : {8, 12}) {
        const uint32_t spg = (kSpans + cus - 1) / cus;
        snprintf(name, sizeof name, "contiguous range per workgroup (168 spans), %d waves", waves);
        row(name, best_us([&] { k_persistent<0, 0><<<cus, waves * 64>>>(buf, spg); }));
        for (uint32_t chunk : {84u, 42u, 21u, 11u, 4u, 1u}) {
          const uint32_t cpg = (kSpans + chunk * cus - 1) / (chunk * cus);
          snprintf(name, sizeof name, "interleaved, chunks of %u spans, %d waves", chunk, waves);
          row(name, best_us([&] { k_interleaved<<<cus, waves * 64>>>(buf, chunk, cpg); }));
        }
      }
    return 0;
  }
  if (getenv("SPLIT")) {
    const uint32_t spg = (kSpans + cus - 1) / cus;
    char name[160];
    for (int waves : {4, 8, 12, 16}) {
      snprintf(name, sizeof name, "split: %d waves, 4 store, the others sleep", waves);
      row(name, best_us([&] { k_split<4, 0><<<cus, waves * 64>>>(buf, spg); }));
      snprintf(name, sizeof name, "split: %d waves, 4 store, the others read LDS flat out", waves);
      row(name, best_us([&] { k_split<4, 1><<<cus, waves * 64>>>(buf, spg); }));
    }
    for (int waves : {8, 12, 16}) {
      snprintf(name, sizeof name, "split: %d waves, 8 store, the others sleep", waves);
      row(name, best_us([&] { k_split<8, 0><<<cus, waves * 64>>>(buf, spg); }));
      snprintf(name, sizeof name, "split: %d waves, 2 store, the others sleep", waves);
      row(name, best_us([&] { k_split<2, 0><<<cus, waves * 64>>>(buf, spg); }));
      snprintf(name, sizeof name, "split: %d waves, 3 store, the others sleep", waves);
      row(name, best_us([&] { k_split<3, 0><<<cus, waves * 64>>>(buf, spg); }));
      snprintf(name, sizeof name, "split: %d waves, 6 store, the others sleep", waves);
      row(name, best_us([&] { k_split<6, 0><<<cus, waves * 64>>>(buf, spg); }));
    }
    return 0;
  }
  if (getenv("QUICK")) {
    persistent_rows<0, 0, 1>(buf, "16 B chunks", "default", cus);
    persistent_rows<0, 0, 2>(buf, "16 B chunks", "default", cus);
    persistent_rows<0, 0, 3>(buf, "16 B chunks", "default", cus);
    persistent_rows<0, 2, 1>(buf, "16 B chunks", "sc1", cus);
    persistent_rows<0, 2>(buf, "16 B chunks", "sc1", cus);
    return 0;
  }
  persistent_rows<0, 0, 1>(buf, "16 B chunks", "default", cus);
  persistent_rows<0, 0, 2>(buf, "16 B chunks", "default", cus);
  persistent_rows<0, 0, 3>(buf, "16 B chunks", "default", cus);
  persistent_rows<0, 1>(buf, "16 B chunks", "nt", cus);
  persistent_rows<0, 2>(buf, "16 B chunks", "sc1", cus);
  persistent_rows<0, 3>(buf, "16 B chunks", "sc0 sc1", cus);
  persistent_rows<0, 4>(buf, "16 B chunks", "sc0 sc1 nt", cus);
  persistent_rows<1, 0>(buf, "12+12 B rows", "default", cus);
  persistent_rows<2, 0>(buf, "dword", "default", cus);
  return 0;
}
