// mp_common.h — shared device/host definitions of the MI355X substrate engine.
//
// HBM layout (one engine = N worlds of one substrate):
//
//   state    u8 [N][world_stride]     one contiguous record per world:
//              [0, grid_bytes)          u8 grid[L][H][W]  state id per cell-layer
//                                       (layer-major planes: SoA inside a world)
//              [grid_pad, +sizeof(WorldTail))  avatars, timers, counters, RNG key
//            world_stride is a multiple of 64 B, so a wavefront streams its
//            world in with 16-byte lane loads (1 KiB per instruction) and the
//            whole record (≈6 KB for clean_up) lives in LDS while it is stepped.
//   tables   the MPK1 pack, copied once; read-only, L2 resident.
//   outputs  caller-owned observation tensors (RGB written with 8/16-byte
//            stores straight from the render kernels) + small f64 arrays.
//
// What the state record restates: the per-piece state/position/orientation the
// reference keeps inside dmlab2d's grid (component_library.lua:51-198 StateManager,
// :211-536 Transform) and the Lua-side volatile variables of the avatar
// components (avatar_library.lua:137-146, :698-707; clean_up/components.lua:234-237).
#ifndef MP_COMMON_H_
#define MP_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mp_engine.h"  // MpEventType, MP_EVENT_ROWS, MP_OBS_*

#define MP_MAX_PLAYERS 16
#define MP_WAVE 64

// Per-world record tail (follows the grid planes).  Plain bytes wherever the
// reference's value range allows it: the tail is read and written once per step.
struct WorldTail {
  uint8_t ax[MP_MAX_PLAYERS];       // Transform position (valid while alive)
  uint8_t ay[MP_MAX_PLAYERS];
  uint8_t aori[MP_MAX_PLAYERS];     // Transform orientation 0..3 = N,E,S,W
  uint8_t aalive[MP_MAX_PLAYERS];   // state == aliveState
  uint8_t ztimer[MP_MAX_PLAYERS];   // Zapper._coolingTimer
  uint8_t ctimer[MP_MAX_PLAYERS];   // substrate aux timer (clean_up: Cleaner)
  uint8_t flag0[MP_MAX_PLAYERS];    // clean_up: GlobalData cleanedThisStep
  uint8_t flag1[MP_MAX_PLAYERS];    // clean_up: GlobalData ateThisStep
  uint8_t freeze[MP_MAX_PLAYERS];   // Avatar._freezeCounter
  uint8_t removal[MP_MAX_PLAYERS];  // Avatar._removalCounter
  uint8_t aflags[MP_MAX_PLAYERS];   // bit0 Avatar._movementAllowed, bit1 Zapper._disallowZapping
  uint8_t nozap[MP_MAX_PLAYERS];    // Zapper._noZappingCounter
  uint8_t level[MP_MAX_PLAYERS];    // GraduatedSanctionsMarking._level
  uint8_t tsince[MP_MAX_PLAYERS];   // GraduatedSanctionsMarking._timeSinceNotInitial
  int32_t achange[MP_MAX_PLAYERS];  // frame of the avatar's last state change
  int32_t step;          // advance() calls this episode
  int32_t frame;         // engine frame counter (grid:update calls)
  int32_t done;          // last advance returned continue == false
  int32_t cont;          // BaseSimulation:continue()
  int32_t aux_count;     // clean_up: RiverMonitor dirtCount
  int32_t group_change;  // clean_up: change frame shared by all water pieces; coins: the colour pair
  uint32_t episode;      // resets so far; episode e draws with counter word 3 = e
  int32_t started;       // 0 until the first reset
  uint64_t seed;         // per-world base seed
  uint32_t ctr[8];       // cumulative counters, see MP_CTR_* (per world)
  int32_t reward_fx;     // cumulative reward, 1/1024 units (signed: coins pays -2)
  // The shuffled visiting orders (A1) of step `orders_step` of this episode, left here by the
  // finish() of the step before it — behind the hand-over to the renderers instead of in front
  // of the next launch's first pixel (stepk::step_orders).  0 = none (steps count from 1);
  // k_set_seeds clears it, a reset passes through finish().  Lane p: nibble g = the avatar
  // that stream g visits p-th.
  uint32_t orders_step;
  uint16_t next_orders[MP_MAX_PLAYERS];
};
static_assert(sizeof(WorldTail) == 400, "WorldTail layout");

// Device views of the pack tables + layout scalars; passed to kernels by value.
struct DevTables {
  int32_t H, W, L, P, nstates, nsprites, topology, max_frames, nact;
  int32_t nfields;                  // raw action fields per avatar (actionOrder), <= 4
  uint32_t field_lo, field_hi;      // their actionSpec min / max, one int8 per field
  int32_t P_pack;                   // players the pack was lowered for (table strides); P <= P_pack
  int32_t avatar_layer, sprite_size;
  int32_t vl, vr, vf, vb;           // egocentric window
  int32_t grid_planes;              // L render planes + substrate-private hidden planes
  int32_t grid_bytes, grid_pad, world_stride;
  int32_t n_spawn, n_init_groups;
  const int32_t* init_spawn_cells;  // initial spawn groups' cells, concatenated
  const int32_t* init_spawn_ptr;    // [n_init_groups + 1]
  const int32_t* avatar_init_group; // [P]
  const uint8_t* init_grid;         // [L][H][W]
  const int32_t* state_layer;       // [nstates]
  const int32_t* state_sprite;      // [nstates]
  const uint8_t* step_blob;         // the step's LDS tables (step_common.h: Tables)
  uint32_t* fault;                  // [16] first pipeline stall of a frame kernel (frame.hip), 0 = none
  uint32_t* claim;                  // [2] the frame kernels' pool counters: a launch counts on one and zeroes the other
                                    // (+ [2 + 2 g]: developer build -DMP_FRAME_ENDS, workgroup g's first pass / end stamps)
  const uint32_t* init_spawn_mask;  // [n_init_groups] group bit of each initial spawn group
  const int32_t* alive_state;       // [P]
  const int32_t* wait_state;        // [P]
  const int32_t* action_table;      // [nact][4]
  const int32_t* spawn_cells;       // [n_spawn] respawn group, y*W+x, creation order
  const int32_t* hit_state;         // [nhits]
  const int32_t* hit_state_dir;     // [nhits][4] beam pseudo-state per beam direction
  const int32_t* state_orient;      // [nstates] facing implied by a pseudo-state
  // renderer
  const uint8_t* sprite_rgba;       // [nsprites][4][S][S][4]
  const int32_t* view_sprite_map;   // [P+1][nsprites]
  const uint8_t* sprite_flags8;     // [nsprites] bit0 opaque, bit1 has 0<alpha<255
  const uint8_t* atlas_compact;     // [n_images][8][8][4] de-duplicated sprite images
  const uint16_t* img_slot;         // [nsprites*4] (sprite, facing) -> image (>= 1)
  int32_t n_images;                 // images in atlas_compact (image 0 is black)
  // composite cache: (opaque image, overlay image drawn directly on it) -> the
  // pre-blended opaque image, for pairs that static pieces of the map can form.
  // Open-addressing table of kPairSlots entries: a << 20 | b << 10 | composite,
  // 0xffffffff = empty; slot = pair_hash(a, b), linear probing, pair_probe max.
  const uint32_t* pair_table;
  int32_t pair_probe;               // 0 = no table
  int32_t scratch_cells;            // composited cells a render wave can stage per pass
  // render planes that can show anything: bit l = a state of plane l has a sprite with a visible
  // pixel, bit 16 + l = a state of it is an avatar's (frame.hip render_visible_layers); the
  // renderers read no other plane
  uint32_t vis_layers;
  // everything the renderer's workgroups stage that does not depend on the
  // world: atlas at LDS stride + lookup tables, laid out exactly as in LDS
  // (render.hip: render_lds_layout, bytes [0, world))
  const uint8_t* render_blob;
  // per-episode 'choice' map characters (prefab_utils.lua:101-103): objects that
  // exist only in some outcomes of their choice — (cell, plane, choice, outcome
  // mask) per object that starts on the grid — and the outcome count per choice
  const uint32_t* state_groups;     // [nstates] group membership bits
  int32_t n_optional;
  int32_t optional_spawn;           // some optional object is a spawn point (spawn_avatars filters)
  const int32_t* optional;          // [n_optional][4]
  const int32_t* choice_n;          // [n_choices]
};

// Geometry of one frame-kernel launch (frame.hip: plan_frame).
struct FramePlan {
  int32_t B;        // worlds per batch
  int32_t NB;       // batches resident in LDS (a ring of NB * B record slots)
  int32_t feeders;  // feeder waves (the last ones of the workgroup)
  int32_t nwaves;   // waves per workgroup
  int32_t groups;   // workgroups (<= CUs)
  int32_t ks;       // batches a workgroup OWNS: batch ids [g * ks, (g + 1) * ks), a contiguous range of worlds
  int32_t pool;     // batches beyond groups * ks, claimed one at a time from a device-wide counter
  int32_t world_waves;   // two views in one launch: renderer waves (the last ones) that draw WORLD.RGB
  int32_t slot_scratch;  // step scratch bytes per feeder slot
  int32_t late_prio;     // wave priority of the feeders once their first world is published
  int32_t parity;        // which of DevTables::claim's two counters this launch counts on
  int32_t store_sc1;     // 1: the pixels leave as sc1 stores (instead of nt in the fused form, plain in the draw-only one)
  int32_t team;          // 1: single-world batches dealt to XCD teams (frame.hip: each XCD writes one compact front)
  int32_t pace;          // what a renderer wave sleeps between two passes, in units of 512 cycles (mp_tune: a launch
                         // that writes faster than the memory side takes its view's pages is SLOWER for it)
  int32_t head;          // the start of a stepping launch (frame.hip): 1 = a feeder's tables and FIRST record
                         // go global -> LDS by DMA, requested before anything else and waited for at its
                         // first step, at feeder priority from its first instruction; 0 = the older road
};

// Beam footprint: cell j of a beam sits `lat` cells to the avatar's right and
// `fwd` cells ahead; bit i of pred[j] = cell i must not stop the beam for cell j
// to be reached (Zapper:getWhoZappable, avatar_library.lua:780-824).
constexpr int kPairSlots = 256;
__host__ __device__ inline uint32_t pair_hash(uint32_t a, uint32_t b) {
  return (((a << 10) | b) * 2654435761u) >> 24;
}

struct BeamShape {
  int32_t n;
  int32_t per;         // 64 / n: beams evaluated per round of lanes
  uint32_t magic;      // 65536 / n + 1: lane / n == (lane * magic) >> 16 for lane < 64
  uint32_t cell[16];   // lat (i8) | fwd (i8) << 8 | pred << 16
};

// Zapper kwargs + where its beam is drawn (avatar_library.lua:570-763).
struct ZapRules {
  int32_t cooldown, length, radius, respawn_frames, remove_hit;
  int32_t layer, s_hit;   // beamZap layer, <hit>.zapHit pseudo-state
  int32_t hit;            // index of zapHit among the pack's hits
  double penalty, reward;
  BeamShape shape;
};

// clean_up rule constants (clean_up.py component kwargs, carried by the pack).
struct CleanUpTables {
  int32_t n_apple, n_dirt, n_water;
  const int32_t* apple_cells;
  const int32_t* dirt_cells;
  const int32_t* water_cells;
  const uint64_t* apple_thr;  // [n_dirt+1] growth threshold by dirt count
  uint64_t thr_dirt_spawn, thr_episode_end;
  int32_t s_apple, s_apple_wait, s_dirt, s_dirt_wait, s_water[4];
  uint32_t s_water_packed;   // s_water[f] << 8 f
  int32_t apple_layer, dirt_layer, dirt_wait_layer, water_layer;
  int32_t clean_layer, s_clean_hit, clean_hit;
  int32_t clean_cooldown, clean_length, clean_radius;
  int32_t dirt_delay, ee_min_frames, ee_interval, anim_frames;
  int32_t n_dirt_init;
  double eat_reward;
  ZapRules zap;
  BeamShape clean_shape;
};

// commons_harvest rule constants (commons_harvest__open.py, in the pack).
struct CommonsTables {
  int32_t n_apple, nk, ndisc;
  const int32_t* apple_cells;
  const int32_t* disc;        // [ndisc][2] queryDisc offsets, self excluded
  const uint64_t* thr;        // [nk] regrowth thresholds, then episode end
  int32_t s_apple, s_wait, s_grass, s_dess, s_wait_k[32];
  int32_t live_layer, wait_layer, grass_layer;
  int32_t ee_min_frames, ee_interval;
  double eat_reward;
  ZapRules zap;
};

// coins rule constants (coins.py, in the pack: one instance of its random map).
struct CoinsTables {
  int32_t n_coin;
  const int32_t* coin_cells;
  int32_t s_coin[2], s_wait, coin_layer, wait_layer;
  int32_t player_type[MP_MAX_PLAYERS];   // PlayerCoinType: index into s_coin
  double rew[2][4];   // per collector: self match / mismatch, others match / mismatch
  uint64_t thr_regrow, thr_ee;
  int32_t ee_min_frames, ee_interval;
  // per-world colours (coins.py:500 draws two of five when an environment is
  // built): the coin's state and each avatar's alive state per colour; 0 = the
  // pack carries one pair only
  // (one byte per colour, selected by shift: indexing a kernel-argument array
  // costs a scratch copy of the struct)
  int32_t has_colours;
  uint64_t colour_coin;        // byte k: the coin's state in colour k
  uint64_t colour_alive[2];    // byte k: avatar p's alive state in colour k
};

// coop_mining rule constants (coop_mining.py, in the pack).  Ore type 0 is the one-miner
// type (iron), type 1 the many-miner type (gold): the Lua indexes its reward tables with
// minNumMiners.
struct CoopTables {
  int32_t n_ore;
  const int32_t* ore_cells;
  const double* reward;       // [P][4]: mining type 0, 1; extracting type 0, 1
  uint64_t thr[3];            // regrow type 0, type 1; episode end
  int32_t s_wait, s_raw[2], s_partial[2];
  int32_t ore_layer;
  int32_t plane_m, plane_c;   // hidden planes: type 1's miners (byte mask), its countdown
  int32_t min_miners1, window1;
  int32_t cooldown, hit, beam_layer, s_beam;
  int32_t ee_min_frames, ee_interval;
  BeamShape shape;
};

// gift_refinements rule constants (gift_refinements.py, in the pack).  An avatar's inventory
// (numTokenTypes <= 3 counts of at most 15) lives in the record's per-avatar bytes
// (WorldTail::flag0 / flag1 / level), its consumption timer in ctimer, the beam's in ztimer.
struct GiftTables {
  int32_t n_token;
  const int32_t* token_cells;
  const double* reward;       // [P][2]: per hit (roleRewardForGifting), per refined gift
  double pick_reward;
  uint64_t thr[2];            // regrow; episode end
  int32_t s_wait, s_live, token_layer;
  int32_t capacity, ntypes, multiplier, consume_cooldown;
  int32_t cooldown, hit, beam_layer, s_beam;
  int32_t ee_min_frames, ee_interval;
  BeamShape shape;
};

// collaborative_cooking rule constants (collaborative_cooking.py, in the pack).  The per-avatar
// interact hits have consecutive layers and beam states, the loading bar's eleven states and
// the inventory's item states are consecutive ids (mp_create checks).
struct CookTables {
  const uint8_t* state_kind;     // [nstates] COOK_KIND_* of an object state (step_cook.h)
  int32_t n_cont, n_pot;
  const int32_t* cont_cells;     // containers (counters, dispensers) in creation order
  const int32_t* cont_i32;       // [n_cont][2]: startingItem, infinite
  const int32_t* pot_cells;
  int32_t s_pot[5];              // a pot holding 0 / 1 / 2 / 3 ingredients, cooked
  int32_t s_bar0;                // loading_bar_0 (.. + 10)
  int32_t s_plain0, s_off0, s_dir0;   // inventory states: item k; item k offset facing N; (facing - 1) * 4 + k
  int32_t overlay_layer, plane_t;
  int32_t beam_layer0, s_beam0;  // avatar p's interact layer / sprite state: + p
  int32_t cooldown, cooking_time, bar_interval;
  int32_t recv_item, recv_global;
  double recv_reward, pot_reward;
};

// externality_mushrooms rule constants (externality_mushrooms.py, in the pack).  The four live
// states of the mushroom prefab are consecutive ids (mp_create checks); `i32` is the pack's
// em_i32 (per type: spores 8.., digestion 12.., typeToDestroy 20..), `thr` its em_thr (grow
// [eaten][grown], then percentToDestroy per type).
struct MushroomTables {
  int32_t n_site, n_live_init;
  const int32_t* site_cells;
  const int32_t* i32;
  const uint64_t* thr;
  int32_t s_type0, live_layer, plane_age, mark_layer;
  int32_t s_mark[2];
  uint32_t perish_packed;             // Perishable delay of type k in byte k, 255 = never
  int32_t min_potential, recovery_time;
  int32_t lv_increment[2], lv_freeze[2], lv_remove[2];
  int32_t ee_min_frames, ee_interval;
  uint64_t thr_ee;
  double rew_self[4], rew_other[4];   // one mushroom of a type: what the eater / every other avatar gets
                                      // (components.lua:65-105 with this engine's player count)
  uint32_t pays;                      // bit k: type k pays the eater; bit 4 + k: it pays the others
  double lv_source[2], lv_target[2];
  ZapRules zap;
};

// territory rule constants (territory.py / territory__rooms.py, in the pack).
struct TerritoryTables {
  int32_t n_res, map_cells;         // resources; H * W
  const int32_t* res_cells;
  int32_t s_res_unclaimed, s_tex_destroyed_unused, s_dmg_inactive, s_dmg_damaged;
  int32_t s_mark[2], s_claimed[MP_MAX_PLAYERS], s_dry[MP_MAX_PLAYERS];
  int32_t res_layer, tex_layer, ind_layer, dmg_layer, mark_layer;
  int32_t brush_layer, claim_layer;            // hit sprite layers
  int32_t plane_a, plane_b, plane_c;           // hidden planes (see step_territory.hip)
  int32_t initial_health, reward_delay, repair_delay;
  int32_t claim_length, claim_wait, recovery_time;
  int32_t lv_increment[2], lv_freeze[2], lv_remove[2];
  int32_t ee_min_frames, ee_interval;
  int32_t hit_zap, hit_brush[MP_MAX_PLAYERS], hit_claim[MP_MAX_PLAYERS];
  int32_t s_brush[MP_MAX_PLAYERS][4], s_claim_hit[MP_MAX_PLAYERS];
  uint64_t thr_reward, thr_repair, thr_ee;
  double reward, lv_source[2], lv_target[2];
  ZapRules zap;
};

// *_in_the_matrix rule constants (<game>_in_the_matrix__<variant>.py + the_matrix.py,
// in the pack: mx_*).  Per-player constants stay in the pack (read per lane from
// global memory once a step); everything here is read with uniform indices only.
struct MatrixTables {
  int32_t n_site;               // site entries: every alternative of a 'choice' cell has one
  const int32_t* site_cells;    // [n_site] y * W + x
  const int32_t* site_class;    // [n_site] 1-based resourceClass
  const int32_t* player_i32;    // [P_pack][4] Taste class, InteractionTaste class, zeroDefault, DyadicRole (-1 none)
  const double* player_f64;     // [P_pack][4] mostTastyReward, defaultTastinessReward, extraReward
  int32_t R;                    // resource classes = strategies
  int32_t res_layer, mark_layer, beam_layer, s_beam, hit;
  int32_t plane_a, plane_b;     // hidden per-cell planes (step_matrix.h)
  int32_t player_block;         // byte offset of MxPlayer[16] in the record
  uint32_t s_visible_packed;    // visible state of class k in byte k (0 = no such class)
  uint64_t s_mark_packed;       // marker state of indicator i (notReady, ready, colour 1..5) in byte i
  int32_t cooldown, respawn_frames, freeze, end_on_first, reset_winner, reset_loser,
      loser_dies, winner_dies, zero_inventory, random_tie, disallow_unready, has_ee,
      ee_min_frames, ee_interval, regen_delay, initial_health, n_intervals, spawn_all;
  double reward_floor, reward_multiplier, reward_unready;
  double row_matrix[9], col_matrix[9];   // [R][R] row-major
  double interval[10];                   // resultIndicatorColorIntervals [n][2]
  uint64_t thr_regen, thr_ee;
  BeamShape shape;
};

// The rule constants of the engine's substrate (one member is in use).
struct SubstrateTables {
  int32_t substrate;   // MPK_SUBSTRATE_*
  CleanUpTables cu;
  CommonsTables ch;
  TerritoryTables tr;
  CoinsTables co;
  MatrixTables mx;
  CoopTables cm;
  GiftTables gr;
  CookTables cc;
  MushroomTables em;
};

// Output pointers for one submission (bound caller buffers or engine-owned).
struct StepOutputs {
  double* reward;        // [N][P]
  double* ready;         // [N][P]
  double* aux0;          // [N][P]
  int32_t* step_type;    // [N]
  double* discount;      // [N]
  double* collective;    // [N]
  int32_t* position;     // [N][P][2]
  int32_t* orientation;  // [N][P]
  int32_t* events;       // [N][MP_EVENT_ROWS][4] (include/mp_engine.h: MP_OBS_EVENTS)
  // debug observations, written only when bound (NULL otherwise)
  double* dbg[4];        // [N][P] each: MP_OBS_AUX1 .. MP_OBS_AUX4
  double* zap_matrix;    // [N][P][P] zapped x zapper, this step
  // *_in_the_matrix (NULL elsewhere)
  double* inventory;     // [N][P][R]     "N.INVENTORY"
  double* interaction;   // [N][P][2][R]  "N.INTERACTION_INVENTORIES"
  double* cumulants;     // [N][P][1 + 3 R] MP_OBS_MATRIX_CUMULANTS (debug: NULL unless bound)
  double* interaction_rewards;   // [N][P][2] MP_OBS_INTERACTION_REWARDS
};

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), the engine's counter-based generator.
// One draw = Philox(counter = {index, stream, step, episode}, key = world seed):
// (world, episode) is an injective key even for worlds with adjacent seeds.
// Replaces the reference's serial mt19937_64 `system.random`
// (api_factory.lua:56,89) — assumption A10 in DESIGN.md.
struct Philox4 { uint32_t x0, x1, x2, x3; };

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1,
                                                 uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
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
  return Philox4{c0, c1, c2, c3};
}

enum {  // streams (counter word 1); same numbering as the CPU restatement
  RS_START_SPAWN = 1, RS_START_ORIENT = 2, RS_ANIM_START = 3,
  RS_APPLE_GROW = 4, RS_DIRT_SPAWN = 5, RS_EPISODE_END = 6,
  RS_SHUFFLE_MOVE = 7, RS_SHUFFLE_ZAP = 8, RS_SHUFFLE_CLEAN = 9,
  RS_SHUFFLE_RESPAWN = 10, RS_RESPAWN = 11, RS_REGROW = 12,
  RS_SHUFFLE_BRUSH = 13, RS_SHUFFLE_CLAIM = 14, RS_RESOURCE_REWARD = 15,
  RS_SELF_REPAIR = 16,
  RS_COIN_CHOICE = 17,
  RS_MAP_CHOICE = 18,
  RS_TIE_BREAK = 19,
  RS_MUSHROOM_GROW = 20, RS_MUSHROOM_DESTROY = 21
};

__host__ __device__ inline uint64_t philox_u53(Philox4 o) {
  return (((uint64_t)o.x1 << 32) | o.x0) >> 11;
}
__host__ __device__ inline uint32_t philox_bounded(Philox4 o, uint32_t n) {
  return (uint32_t)(((uint64_t)o.x2 * n) >> 32);
}

// STEP: `actions` are ids into ACTION_SET [N][P]; FIELDS: raw action fields
// [N][P][nfields] in actionOrder (mp_step_fields)
enum { STEP_MODE_STEP = 0, STEP_MODE_RESET = 1, STEP_MODE_FIELDS = 2 };


#endif  // MP_COMMON_H_
