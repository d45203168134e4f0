/* ORACLE — test infrastructure only.  Not linked into the product.
 *
 * Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3",
 * SC'11; Random123 reference constants) and the draw conventions of the
 * engine (assumption A10 in DESIGN.md: a counter-based generator replaces the
 * reference's serial mt19937_64 `system.random`, seeded at
 * lua/modules/api_factory.lua:56,89 — per-draw values differ from DMLab2D by
 * design, the *distributions* follow the reference call sites).
 *
 * One draw = Philox(counter = {index, stream, step, 0}, key = episode seed):
 *   u53     = ((x1 << 32 | x0) >> 11)           uniformReal(0,1) = u53 * 2^-53
 *   bounded = (x2 * n) >> 32                    random:choice / shuffles
 *   x3 & 3                                      orientation picks
 */
#ifndef ORACLE_PHILOX_H_
#define ORACLE_PHILOX_H_
#include <stdint.h>

typedef struct { uint32_t x[4]; } PhiloxOut;

static inline PhiloxOut philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                      uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  PhiloxOut o = {{c0, c1, c2, c3}};
  return o;
}

/* streams (counter word 1) */
enum {
  RS_START_SPAWN = 1,   /* groupShuffledWithCount, base_simulation.lua:418 */
  RS_START_ORIENT = 2,  /* random:choice(_COMPASS), avatar_library.lua:301 */
  RS_ANIM_START = 3,    /* Animation:postStart, component_library.lua:1064 */
  RS_APPLE_GROW = 4,    /* AppleGrow:update, clean_up/components.lua:77 */
  RS_DIRT_SPAWN = 5,    /* DirtSpawner:update, clean_up/components.lua:331 */
  RS_EPISODE_END = 6,   /* component_library.lua:930 */
  RS_SHUFFLE_MOVE = 7,  /* engine: piece order inside updater 150 */
  RS_SHUFFLE_ZAP = 8,   /* engine: piece order inside updater 140 (zap) */
  RS_SHUFFLE_CLEAN = 9, /* engine: piece order inside updater 140 (clean) */
  RS_SHUFFLE_RESPAWN = 10, /* engine: piece order inside updater 135 */
  RS_RESPAWN = 11,      /* teleportToGroup target + PICK_RANDOM orientation */
  RS_REGROW = 12,       /* engine: probabilistic updater on the waits_k groups
                           (commons_harvest/components.lua:119-136) */
  RS_SHUFFLE_BRUSH = 13,  /* engine: piece order inside updater 130 (Paintbrush) */
  RS_SHUFFLE_CLAIM = 14,  /* engine: piece order inside updater 100 (ResourceClaimer) */
  RS_RESOURCE_REWARD = 15, /* engine: probabilistic provideRewards updater
                              (territory/components.lua:85-102) */
  RS_SELF_REPAIR = 16,  /* Resource:update, territory/components.lua:197 */
  RS_COIN_CHOICE = 17,  /* random:choice(liveStates), coins/components.lua:198 */
  RS_MAP_CHOICE = 18,   /* random:choice(prefab.list) at world build, prefab_utils.lua:101-103 */
  RS_TIE_BREAK = 19,    /* randomTieBreaking, the_matrix/components.lua:614-621 (index = zapped player) */
  RS_MUSHROOM_GROW = 20,    /* MushroomRegrowth:grow, externality_mushrooms/components.lua:220-222
                               (index = ((eater * 4 + spore) * 4 + type): the uniform and the choice) */
  RS_MUSHROOM_DESTROY = 21  /* getGroupShuffledWithProbability, :237-244 (index = eater * 256 + site) */
};

static inline uint64_t philox_u53(PhiloxOut o) {
  return (((uint64_t)o.x[1] << 32) | o.x[0]) >> 11;
}
static inline uint32_t philox_bounded(PhiloxOut o, uint32_t n) {
  return (uint32_t)(((uint64_t)o.x[2] * n) >> 32);
}
#endif
