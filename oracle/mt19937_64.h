/* ORACLE — test infrastructure only.  Not linked into the product.
 *
 * mt19937_64 (Matsumoto & Nishimura, "Mersenne Twister", the 64-bit variant of
 * 2004: w = 64, n = 312, m = 156, r = 31, a = 0xB5026F5AA96619E9, the tempering of
 * std::mt19937_64) — the reference's ONE serial generator: `system.random`, seeded by
 * `random:seed(seed)` in api:init / api:start (lua/modules/api_factory.lua:56,89) and
 * consumed by every Lua draw (clean_up/components.lua:77,331-332; base_simulation.lua:
 * 418; component_library.lua:930 ...) and by the engine's own shuffles
 * (`self._grid:update(random)`, api_factory.lua:101,106).
 *
 * Known answer (the C++ standard's, [rand.predef]): the 10000th consecutive
 * invocation of a default-constructed std::mt19937_64 (seed 5489) produces
 * 9981545732273789042 (tests/test_oracle_cpu.py).
 *
 * The oracle's default generator stays the counter-based one (assumption A10); this
 * one is the switchable alternative A10s (`orc_set_option` 6), with the conversions
 * from a 64-bit output to a real / a bounded integer restated below as further
 * switchable assumptions — which of them DMLab2D's build uses is decided by a real
 * trace.
 */
#ifndef ORACLE_MT19937_64_H_
#define ORACLE_MT19937_64_H_
#include <stdint.h>

typedef struct { uint64_t mt[312]; int idx; } Mt64;

static inline void mt64_seed(Mt64* g, uint64_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 312; ++i)
    g->mt[i] = 6364136223846793005ull * (g->mt[i - 1] ^ (g->mt[i - 1] >> 62)) + (uint64_t)i;
  g->idx = 312;
}

static inline uint64_t mt64_next(Mt64* g) {
  if (g->idx >= 312) {
    for (int i = 0; i < 312; ++i) {
      uint64_t x = (g->mt[i] & 0xFFFFFFFF80000000ull) | (g->mt[(i + 1) % 312] & 0x7FFFFFFFull);
      uint64_t xa = x >> 1;
      if (x & 1ull) xa ^= 0xB5026F5AA96619E9ull;
      g->mt[i] = g->mt[(i + 156) % 312] ^ xa;
    }
    g->idx = 0;
  }
  uint64_t y = g->mt[g->idx++];
  y ^= (y >> 29) & 0x5555555555555555ull;
  y ^= (y << 17) & 0x71D67FFFEDA60000ull;
  y ^= (y << 37) & 0xFFF7EEE000000000ull;
  y ^= y >> 43;
  return y;
}

/* uniformReal(0, 1) as a 53-bit integer u (the value is u * 2^-53), so that the
 * oracle's threshold compares stay integer compares.  std::uniform_real_distribution
 * <double> on a 64-bit engine is generate_canonical<double, 53>: ONE output x,
 * converted to double (round to nearest even at 53 bits) and divided by 2^64; a
 * result of 1.0 is replaced by the largest double below it (libstdc++
 * bits/random.tcc, libc++ __generate_canonical alike). */
static inline uint64_t mt64_u53(Mt64* g) {
  const uint64_t x = mt64_next(g);
  uint64_t u = x >> 11;
  const uint64_t rest = x & 0x7FFull;
  if (rest > 0x400ull || (rest == 0x400ull && (u & 1ull))) ++u;   /* nearest, ties to even */
  if (u >> 53) u = (1ull << 53) - 1ull;
  return u;
}

/* uniformInt(0, n - 1): std::uniform_int_distribution<uint64_t>.
 * method 0 — libstdc++ since GCC 11 (Lemire's nearly divisionless method, 128-bit
 *   product, rejection below (2^64 - n) % n);
 * method 1 — libstdc++ before GCC 11 and its generic path (scaling = 2^64 / n by
 *   the (max - min) / range form, rejection at and above n * scaling, then x / scaling). */
static inline uint64_t mt64_bounded(Mt64* g, uint64_t n, int method) {
  if (n <= 1) return 0;   /* a == b: the distribution does not call the engine */
  if (method == 0) {
    unsigned __int128 product = (unsigned __int128)mt64_next(g) * n;
    uint64_t low = (uint64_t)product;
    if (low < n) {
      const uint64_t threshold = (0ull - n) % n;
      while (low < threshold) {
        product = (unsigned __int128)mt64_next(g) * n;
        low = (uint64_t)product;
      }
    }
    return (uint64_t)(product >> 64);
  }
  const uint64_t scaling = 0xFFFFFFFFFFFFFFFFull / n, past = n * scaling;
  uint64_t x;
  do x = mt64_next(g); while (x >= past);
  return x / scaling;
}
#endif
