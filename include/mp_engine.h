/* mp_engine.h — C ABI of the MI355X batched substrate engine (libmp_engine.so).
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json: the
 * DMLab2D/Lua step + render path of Melting Pot.  In the reference that path
 * sits behind the `dmlab2d.Environment` object built at
 * meltingpot/utils/substrates/builder.py:179-187 and driven through
 * meltingpot/utils/substrates/wrappers/base.py:38-84
 * (reset / step / observation / events / *_spec / close).  Underneath, dmlab2d
 * drives the Lua API object of lua/modules/api_factory.lua:26-115
 * (init / start / discreteActions / advance / observation).  Each entry point
 * below names the reference interface it replaces.  INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - Plain C, no torch / HIP types in the signatures (a stream is a void*
 *     holding a hipStream_t; device buffers are raw device pointers).
 *   - An engine owns N independent worlds of one substrate on one GPU.  The
 *     caller owns every action / observation buffer it passes in
 *     (`tensor.data_ptr()`); the engine never frees or retains caller memory
 *     beyond what mp_bind_output documents.
 *   - Every function returns MP_OK (0) or a negative MP_ERR_* code;
 *     mp_last_error() returns a message for the calling thread.
 *   - Like a Lab2d instance an engine is not re-entrant: one host thread per
 *     engine.  All work is enqueued on the engine's stream (mp_set_stream);
 *     mp_step / mp_observe never synchronise the host.
 *   - There is NO CPU fallback: every entry point that needs a GPU fails with
 *     MP_ERR_NO_DEVICE when none is present.
 */
#ifndef MP_ENGINE_H_
#define MP_ENGINE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden and linked against an export map
 * (meltingpot_amd/csrc/exports.map): what this header declares is ALL it exports. */
#pragma GCC visibility push(default)

/* 8: mp_place_output_ring and mp_alloc_output_scattered are gone (measured: they did not pay);
 * mp_box_fill — what the box's memory system gives the bound view — is new; MpInfo.plan_pace /
 * visible_layers / plan_team / plan_late_priority */
#define MP_ABI_VERSION 8

enum {
  MP_OK = 0,
  MP_ERR_INVALID = -1,    /* bad argument (the reference raises ValueError) */
  MP_ERR_PACK = -2,       /* malformed / unsupported substrate pack */
  MP_ERR_NO_DEVICE = -3,  /* no HIP device: the engine has no CPU path */
  MP_ERR_HIP = -4,        /* a HIP runtime call failed */
  MP_ERR_UNSUPPORTED = -5 /* observation not provided by this substrate */
};

/* Observation kinds (reference names: clean_up.py:813-832, specs.py:26-43,
 * avatar_library.lua:225-277,869-881, component_library.lua:786-803). */
/* The events the three levels emit on the hot path (Lua `events:add(name,
 * 'dict', key, value, ...)`), payload ints a, b; player indices are 1-based as
 * in Lua.  Rows of one step are in no particular order: sort them. */
typedef enum {
  MP_EVENT_ZAP = 1,                /* avatar_library.lua:661  a=source b=target */
  MP_EVENT_EDIBLE_CONSUMED = 2,    /* component_library.lua:996, clean_up/components.lua:402  a=player_index */
  MP_EVENT_PLAYER_CLEANED = 3,     /* clean_up/components.lua:152  a=player_index */
  MP_EVENT_CLAIMED_RESOURCE = 4,   /* territory/components.lua:133  a=player_index */
  MP_EVENT_DESTROYED_RESOURCE = 5, /* territory/components.lua:168  a=player_index */
  MP_EVENT_SANCTIONING = 6,        /* avatar_library.lua:1088  a=source b=target */
  MP_EVENT_REMOVAL_DUE_TO_SANCTIONING = 7, /* avatar_library.lua:1070  a=source b=target */
  MP_EVENT_SET_SANCTIONING_LEVEL = 8,      /* avatar_library.lua:1118  a=player_index b=level */
  MP_EVENT_AVATAR_STARTED = 9,     /* avatar_library.lua:317 ('str', 'success'), once per avatar at reset */
  MP_EVENT_COIN_CONSUMED = 10,     /* coins/components.lua:151-154  a=player_index b=player_coin_type << 1 | coin_type
                                      (indices of the level's two coin colours instead of their names) */
  MP_EVENT_INTERACTION = 11,       /* the_matrix/components.lua:790  a=row_player_idx b=col_player_idx
                                      (rewards and inventories: MP_OBS_INTERACTION_INVENTORIES, MP_OBS_REWARD) */
  MP_EVENT_COLLECTED_RESOURCE = 12,/* the_matrix/components.lua:117  a=player_index b=class
                                      (the_matrix's destroyed_resource, :178, is MP_EVENT_DESTROYED_RESOURCE
                                      with b=class) */
  MP_EVENT_MINING = 13,            /* coop_mining/components.lua:196  a=player b=ore_type (1 iron, 2 gold) */
  MP_EVENT_EXTRACTION = 14,        /* coop_mining/components.lua:210  a=player b=ore_type */
  MP_EVENT_EXTRACTION_PAIR = 15,   /* coop_mining/components.lua:220  a=player_a b=player_b << 2 | ore_type */
  MP_EVENT_RECEIVER_ACCEPTED_ITEM = 17,   /* collaborative_cooking/components.lua:325-328  a=player_index
                                             b=item (1 tomato, 2 dish, 3 soup); 'receiver' is the
                                             component's name, "Receiver" */
  MP_EVENT_ITEM_DROPPED_INTO_POT = 18,    /* :397-400  a=player_index b=item; 'pot' = "CookingPot" */
  MP_EVENT_COOKED_FOOD_COLLECTED = 19,    /* :412-415  a=player_index b=cooked_item (3 soup) */
  MP_EVENT_EATING_MUSHROOM = 20,   /* externality_mushrooms/components.lua:72-74  a=player_index b=mushroom_type
                                      (1 fullInternalityZeroExternality, 2 halfInternalityHalfExternality,
                                      3 zeroInternalityFullExternality, 4 negativeInternalityNegativeExternality) */
  MP_EVENT_GIFT = 16               /* gift_refinements/components.lua:176-182  a=gifter_index | source_type << 4
                                      b=receipient_index | received_amount << 4 (the count the recipient
                                      then holds: what Inventory:addTokens returns); the two roles are the
                                      avatars' agentRole kwargs, known to the host */
} MpEventType;
#define MP_EVENT_ROWS 128  /* 1 header row + up to 127 events per world-step; more are counted
                              in the header's `dropped` (never seen: 16 commons_harvest players
                              half of whose actions are beams peak at 16 events in a step over 400 steps,
                              tests/test_gpu_surface.py::test_event_rows_hold_a_zap_storm) */

typedef enum {
  MP_OBS_RGB = 0,            /* "N.RGB"        u8  [N][P][VH*S][VW*S][3] */
  MP_OBS_WORLD_RGB = 1,      /* "WORLD.RGB"    u8  [N][H*S][W*S][3] */
  MP_OBS_REWARD = 2,         /* "N.REWARD"     f64 [N][P] */
  MP_OBS_READY_TO_SHOOT = 3, /* "N.READY_TO_SHOOT" f64 [N][P] */
  MP_OBS_AUX0 = 4,           /* substrate metric 0, f64 [N][P]
                                clean_up: NUM_OTHERS_WHO_CLEANED_THIS_STEP */
  MP_OBS_STEP_TYPE = 5,      /* dm_env.StepType i32 [N]: 0 FIRST 1 MID 2 LAST */
  MP_OBS_DISCOUNT = 6,       /* f64 [N]  (0 on FIRST/LAST, 1 on MID) */
  MP_OBS_COLLECTIVE_REWARD = 7, /* f64 [N] = sum_p REWARD
                                (collective_reward_wrapper.py:49) */
  MP_OBS_POSITION = 8,       /* "N.POSITION" i32 [N][P][2] (x, y); debug obs
                                (avatar_library.lua:806-855) */
  MP_OBS_ORIENTATION = 9,    /* "N.ORIENTATION" i32 [N][P] */
  MP_OBS_EVENTS = 10,        /* env.events() of the last step / reset, i32
                                [N][MP_EVENT_ROWS][4]: row 0 = {count, dropped,
                                0, 0}, rows 1..count = {MpEventType, a, b, 0}
                                (wrappers/base.py:72-74, `events:add` sites
                                listed at MpEventType) */
  /* Debug observations (the reference builds them when a config sets
   * _ENABLE_DEBUG_OBSERVATIONS, clean_up.py:751-784).  They are produced only
   * while a buffer is bound (mp_bind_output) or MpConfig.debug_observations is
   * set; otherwise mp_observe returns MP_ERR_UNSUPPORTED for them. */
  MP_OBS_AUX1 = 11,          /* f64 [N][P]  clean_up: PLAYER_CLEANED
                                (clean_up/components.lua:227,249) */
  MP_OBS_AUX2 = 12,          /* f64 [N][P]  clean_up: PLAYER_ATE_APPLE (:429,458) */
  MP_OBS_AUX3 = 13,          /* f64 [N][P]  clean_up: NUM_OTHERS_PLAYER_ZAPPED_THIS_STEP
                                (avatar_library.lua:672-677) */
  MP_OBS_AUX4 = 14,          /* f64 [N][P]  clean_up: NUM_OTHERS_WHO_ATE_THIS_STEP
                                (clean_up/components.lua:538) */
  MP_OBS_ZAP_MATRIX = 15,    /* f64 [N][P][P]  playerZapMatrix(zapped, zapper) of the
                                step (avatar_library.lua:657-659; GlobalMetricHolder
                                clears it every step, component_library.lua:717-722) */
  MP_OBS_LAYER = 16,         /* "N.LAYER" i32 [N][P][VH][VW][L]: the player's layer
                                view with orientation 'N' (the window is not turned
                                with the avatar, avatar_library.lua:246-257): 1 +
                                sprite index (after the viewer's spriteMap) of the
                                piece or beam in each cell-layer, 0 = nothing;
                                cells outside the map hold OutOfBounds in every
                                layer.  Bound, it is refreshed by
                                one more small launch per step. */
  MP_OBS_INVENTORY = 17,     /* "N.INVENTORY" f64 [N][P][R]: TheMatrix.playerResources
                                (the_matrix/components.lua:942-963); gift_refinements:
                                Inventory.inventory (gift_refinements/components.lua:239-353);
                                R = MpInfo.num_resources */
  MP_OBS_INTERACTION_INVENTORIES = 18, /* "N.INTERACTION_INVENTORIES" f64 [N][P][2][R]:
                                (own, partner's) inventory of the interaction resolved
                                this step, -1 otherwise (the_matrix/components.lua:761-783,
                                899-903) */
  MP_OBS_MATRIX_CUMULANTS = 19, /* *_in_the_matrix debug observations (the_matrix.py:22-60;
                                GameInteractionZapper's binary cumulants, components.lua:
                                808-853), f64 [N][P][1 + 3 R], 0 / 1, columns
                                  0           "N.INTERACTED_THIS_STEP"
                                  1 + 3 k     "N.COLLECTED_RESOURCE_<k+1>"
                                  2 + 3 k     "N.DESTROYED_RESOURCE_<k+1>"
                                  3 + 3 k     "N.ARGMAX_INTERACTION_INVENTORY_WAS_<k+1>"
                                produced while bound or with MpConfig.debug_observations */
  MP_OBS_INTERACTION_REWARDS = 20, /* *_in_the_matrix: f64 [N][P][2], (row_reward,
                                col_reward) of the latest interaction player p took part
                                in — with MP_OBS_INTERACTION_INVENTORIES the rest of the
                                reference's 'interaction' event payload (the_matrix/
                                components.lua:789-797: row_reward, col_reward,
                                row_inventory, col_inventory), exact f64: an event row
                                {MP_EVENT_INTERACTION, row, col} of a step says which
                                players' entries are this step's */
  MP_OBS_KINDS = 21
} MpObsKind;

typedef struct MpEngine MpEngine;

/* Test / development overrides of the launch plan (MpConfig.dev; NULL in every
 * product path: meltingpot_amd.substrate / lab2d_env never set it).  They change
 * how the work is laid out over workgroups and LDS, never a result: the tests
 * use them to drive the frame kernel through its corner geometries (tiny and
 * ragged batches, many batches per workgroup, the direct-store path) and
 * tools/ to sweep the plan.  The library reads NO environment variable.
 * 0 (or -1 for max_composites) = the engine's own choice. */
typedef struct {
  uint32_t struct_size;     /* = sizeof(MpDevOptions) */
  int32_t batch_worlds;     /* worlds per LDS batch of the frame kernel */
  int32_t waves;            /* waves per workgroup */
  int32_t feeders;          /* feeder waves among them */
  int32_t max_groups;       /* cap on workgroups (more batches per workgroup) */
  int32_t scratch_cells;    /* composited cells a wave stages per pass */
  int32_t no_composite_cache; /* 1: every overlay is composited on the fly */
  int32_t max_composites;   /* cap on composite-cache images, -1 = none */
  int32_t verbose;          /* 1: print the plans to stderr */
  int32_t late_feeder_prio; /* 1 + wave priority (0..3) of the feeders after their first batch */
  int32_t ring_batches;     /* batches resident in LDS (the ring's depth), >= 2 */
  int32_t static_pct;       /* 1..100: share of a workgroup's even split it owns as a
                               contiguous range; the rest of the launch's batches are
                               claimed from a device-wide pool (100: no pool) */
  int32_t world_waves;      /* two views in one launch: renderer waves that draw WORLD.RGB */
  int32_t store_sc1;        /* 1: the pixels leave as sc1 stores */
  int32_t head;             /* 1 + FramePlan::head (how a stepping launch starts: 2 = the DMA
                               head, 1 = the older road) */
  int32_t no_next_orders;   /* 1: a step does not leave the NEXT step's shuffled visiting orders
                               in the world's record (it draws them at its own start instead) */
  int32_t record_pad;       /* unused 64-byte blocks behind every world's record (another stride) */
  int32_t pace;             /* 1 + FramePlan::pace: what a renderer wave sleeps between two passes,
                               in units of 512 cycles (mp_tune's throttle for a view the memory side
                               serves unevenly) */
  int32_t team;             /* 1: with single-world batches (batch_worlds = 1, nothing pooled), the workgroups
                               of an XCD share one contiguous range of worlds and deal it among themselves */
} MpDevOptions;

typedef struct {
  uint32_t struct_size;  /* = sizeof(MpConfig) */
  int32_t device;        /* HIP device ordinal */
  int32_t num_worlds;    /* N worlds owned by this engine */
  int32_t auto_reset;    /* 1: a world whose episode ended restarts on the next
                            mp_step (dm_env: step after LAST == reset) */
  uint64_t world_offset; /* global index of this engine's world 0; world w is
                            seeded from (world_offset + w) so results do not
                            depend on how worlds are sharded over GPUs */
  uint64_t base_seed;    /* 0: seed_w = 0x9E3779B97F4A7C15 * (w+1) (BASELINE.md
                            §4); else seed_w = base_seed + w (builder.py:174-181
                            with one env_seed per world) */
  void* stream;          /* hipStream_t, or NULL for the legacy default stream */
  int32_t num_players;   /* 0: the pack's default (MPK_HDR_DEFAULT_P, else all the
                            avatars it was lowered for); else 1 <= num_players <=
                            the pack's count: the first num_players avatars play
                            (the reference: num_players = len(roles),
                            configs/substrates/clean_up.py:847) */
  int32_t debug_observations; /* 1: the engine keeps buffers for the debug
                            observation kinds (MP_OBS_AUX1..) and fills them every step */
  int32_t unfused;       /* launches of a step with a bound RGB view — same results
                            either way.  2: ONE launch (rules and pixels fused);
                            1: one launch for the rules and one per view;
                            0: the engine's choice (the fused launch; DESIGN.md
                            section 3).  MpInfo.fused reports the launch form */
  int32_t literal_base_seed; /* 1: seed_w = base_seed + w even for base_seed 0 (an
                            env_seed of 0 is a seed like any other, builder.py:174-181) */
  const MpDevOptions* dev; /* NULL (product); tests / tools: see MpDevOptions */
  const int32_t* roles;  /* NULL: the assignment the pack was lowered for (the config's
                            default_player_roles).  Else HOST int32[num_players]:
                            the role of each player as an index into the pack's
                            "role_names" (sorted names, NUL-separated) — for
                            substrates whose config builds per-player constants from
                            the roles (bach_or_stravinsky_in_the_matrix__*: row /
                            column player and avatar colour, configs/substrates/
                            bach_or_stravinsky_in_the_matrix__repeated.py:473-497);
                            MP_ERR_INVALID for a pack without per-role tables */
} MpConfig;

typedef struct {
  int32_t abi_version;
  int32_t substrate;     /* MPK_SUBSTRATE_* */
  int32_t num_worlds, num_players, num_actions;
  int32_t map_h, map_w, num_layers, sprite_size;
  int32_t view_h, view_w; /* egocentric window in cells */
  int32_t max_frames;
  int32_t world_state_bytes; /* bytes of HBM-resident state per world */
  int32_t fused;         /* 1: a step with a bound view is one launch (MpConfig.unfused) */
  int32_t num_resources; /* *_in_the_matrix: resource classes R; gift_refinements: token types
                            (0 elsewhere) */
  int32_t num_action_fields; /* A = len(actionOrder): the raw fields of mp_step_fields */
  /* the launch plan of a step with the pixel views bound right now (mp_tune may have
   * replaced the stock one): worlds per LDS batch, batches resident, batches a
   * workgroup owns, batches pooled behind the claim counter, workgroups */
  int32_t plan_batch_worlds, plan_ring_batches, plan_owned_batches, plan_pooled_batches,
          plan_groups, plan_store_sc1 /* 1: sc1 pixel stores */;
  int32_t plan_feeders, plan_waves; /* (ABI 5) feeder waves among the waves of a workgroup */
  /* (ABI 6) the rollout ring (mp_bind_output_ring): its slots (0: none) and the slot the
   * NEXT mp_reset / mp_step writes; the last one written is (ring_next + ring_slots - 1)
   * % ring_slots once anything has been submitted */
  int32_t ring_slots, ring_next;
  /* (ABI 8) what a renderer wave sleeps between two passes under the plan above, in units of 512
   * cycles (mp_tune: 0 on a view the memory side takes evenly), and the render planes that can show
   * anything (bit l: some state of layer l has a sprite with a visible pixel — the only planes the
   * renderers read) */
  int32_t plan_pace, visible_layers;
  /* (ABI 8) 1: the plan deals its single-world batches to XCD teams (each XCD writes one compact
   * front); the feeders' wave priority (0 - 3) after their first world */
  int32_t plan_team, plan_late_priority;
  /* (ABI 6) virtual address space this PROCESS has retired with mapped views
   * (mp_free_output / mp_place_output keep a released view's range reserved), and the
   * bound beyond which mp_alloc_output / mp_place_output refuse to map more */
  int64_t retired_va_bytes, retired_va_limit;
} MpInfo;

/* ABI version of the loaded library. */
int mp_abi_version(void);

/* Message for the last error on this thread ("" if none). */
const char* mp_last_error(void);

/* Construction.  Replaces dmlab2d.Lab2d(root, settings) +
 * dmlab2d.Environment(env, names, seed) (builder.py:182-187) and Lua api:init
 * (api_factory.lua:53-67).  `pack` is an MPK1 blob (include/mp_pack.h): the
 * lowered form of the settings dict the reference passes to builder.builder().
 * All worlds start un-reset; call mp_reset before the first mp_step. */
int mp_create(const void* pack, uint64_t pack_len, const MpConfig* cfg,
              MpEngine** out);

/* dmlab2d.Environment.close() (wrappers/base.py:76-78). */
void mp_destroy(MpEngine* eng);

int mp_info(const MpEngine* eng, MpInfo* out);

/* Work submitted after this call is enqueued on `stream` (a hipStream_t); it is
 * ordered after everything the engine has enqueued on its previous stream. */
int mp_set_stream(MpEngine* eng, void* stream);

/* Register a caller-owned DEVICE buffer for an observation kind (NULL
 * unbinds).  While bound, every mp_reset / mp_step refreshes the buffer as
 * part of the same submission (RGB kinds are rendered straight into it; the
 * scalar kinds are written by the step kernel).  The buffer must stay valid
 * until unbound or mp_destroy.  Replaces the per-name api:observation(idx)
 * reads after each step (api_factory.lua:73-75). */
int mp_bind_output(MpEngine* eng, MpObsKind kind, void* device_ptr);
/* (MP_ERR_INVALID for a pointer the device cannot write — plain host memory, memory of
 * another device: a launch writing through it would fault the GPU in the middle of a step) */

/* A rollout ring: observations a learner keeps without copying them.  The reference
 * hands back FRESH arrays every step (wrappers/multiplayer_wrapper.py:108-118,
 * utils/substrates/substrate.py:74-81) and a rollout simply stores them; a bound
 * buffer (mp_bind_output) is overwritten in place.  With a ring bound for `kind`,
 * submission number t since the ring was bound (every mp_reset and every mp_step* is one
 * submission; binding the first kind of a ring restarts the count) writes the kind
 * into slot t % slots: DEVICE memory `base` + slot * slot_stride_bytes,
 * slot_stride_bytes >= mp_obs_bytes(kind) and a multiple of 256.  All kinds bound as
 * rings share one slot count and one position, so slot s of every kind holds the same
 * step; kinds bound with mp_bind_output keep being overwritten in place.  Rebinding is
 * a pointer store per kind; no call synchronises or re-tunes between steps: mp_tune
 * (once, after binding) times the candidate launch plans on EVERY slot of the bound
 * pixel views and remembers a plan per slot.  mp_observe of a ring-bound scalar kind
 * reads the slot written last.  base == NULL unbinds the kind (like mp_bind_output
 * with NULL; so does mp_bind_output on the kind).  MpInfo.ring_slots / ring_next
 * report the position. */
int mp_bind_output_ring(MpEngine* eng, MpObsKind kind, void* base,
                        uint64_t slot_stride_bytes, int32_t slots);

/* Episode start for the worlds selected by `mask` (HOST u8[N], NULL = all).
 * `seeds` (HOST u64[N], NULL = keep) overrides the per-world seed and restarts
 * the world's episode count.  Episode e of a world draws from the counter-based
 * generator keyed by the world's seed with e in the counter:
 * like the reference's rebuild-with-seed+1 convention (builder.py:177-181,
 * reset_wrapper.py:37-45) every episode has its own stream, and — unlike
 * seed + e — worlds with adjacent seeds never share one.  Replaces
 * api:start(episode, seed) (api_factory.lua:85-102). */
int mp_reset(MpEngine* eng, const uint64_t* seeds, const uint8_t* mask);

/* One environment step for all N worlds.  `actions` is a DEVICE int32[N][P]
 * of discrete action ids into the substrate's ACTION_SET (clean_up.py:473-483;
 * the table lookup of discrete_action_wrapper.py:97-109 happens on device).
 * Out-of-range ids are treated as NOOP and counted in mp_counters[MP_CTR_BAD_ACTIONS].
 * Replaces api:discreteActions + api:advance (api_factory.lua:81,104-111). */
int mp_step(MpEngine* eng, const int32_t* actions_device);

/* Same with a HOST int32[N][P]; validates ids (MP_ERR_INVALID, like
 * discrete_action_wrapper.py:28-49).  The array is copied into a ring of
 * pinned, device-mapped buffers that the step kernel reads directly, so the
 * caller may reuse `actions_host` as soon as the call returns and nothing
 * synchronises the stream (the 4th later call waits for this one's step). */
int mp_step_host(MpEngine* eng, const int32_t* actions_host);

/* The raw action surface of dmlab2d.Environment.step: one int per field of the
 * avatar's actionOrder ("<player>.move", ".turn", ".fireZap" ...;
 * avatar_library.lua:205-223, wrappers/base.py:38-44), DEVICE int32 [N][P][A],
 * A = MpInfo.num_action_fields, any combination inside the actionSpec ranges of
 * the pack's "action_spec" table (move + turn + zap in one step: a scenario's or
 * a human player's action, human_players/level_playing_utils.py:283,333-334;
 * or the rows of a custom `action_table`, discrete_action_wrapper.py:77-109).
 * An avatar with a field outside its range does NOOP and is counted in
 * MP_CTR_BAD_ACTIONS.  mp_step(ids) == mp_step_fields(ACTION_SET[ids]). */
int mp_step_fields(MpEngine* eng, const int32_t* fields_device);

/* Same with a HOST int32 [N][P][A]; validates the ranges (MP_ERR_INVALID). */
int mp_step_fields_host(MpEngine* eng, const int32_t* fields_host);

/* Write observation `kind` for all worlds into the caller-owned DEVICE buffer
 * `dst` (layouts in MpObsKind).  Replaces api:observation(idx). */
int mp_observe(MpEngine* eng, MpObsKind kind, void* dst_device);

/* Bytes of observation `kind` for all N worlds (0 if unsupported). */
uint64_t mp_obs_bytes(const MpEngine* eng, MpObsKind kind);

/* Canonical state dump to HOST buffers (synchronises): the layout the parity
 * tests compare bit-for-bit with the oracle's:
 *   grid u8 [N][L][H][W]   state id of the piece (or beam pseudo-state)
 *   avat i32[N][P][8]      x, y, orient, alive, zap_timer, aux_timer,
 *                          frames_in_state, 0
 *   glob i32[N][8]         step, done, frame, aux_count, episode, 0, 0, 0 */
int mp_dump(MpEngine* eng, uint8_t* grid, int32_t* avat, int32_t* glob);

/* Checkpoint / restore of the raw HBM state of all worlds (synchronises).
 * `bytes` must equal mp_snapshot_bytes().  (The reference has no equivalent:
 * SURVEY.md §5 "checkpoint / resume".) */
uint64_t mp_snapshot_bytes(const MpEngine* eng);
int mp_snapshot(MpEngine* eng, void* host_buf, uint64_t bytes);
int mp_restore(MpEngine* eng, const void* host_buf, uint64_t bytes);

/* Throughput / event counters accumulated on device since creation
 * (synchronises).  These are what the multi-GPU bench all-reduces. */
enum {
  MP_CTR_WORLD_STEPS = 0, MP_CTR_AGENT_STEPS, MP_CTR_EPISODES,
  MP_CTR_REWARD_SUM /* in 1/1024 reward units */, MP_CTR_ZAPS, MP_CTR_AUX0
  /* clean_up: cleans; externality_mushrooms: (marking, frame) pairs in which a sanctions marking
     was on the map away from its living avatar (avatar_library.lua:1099-1110: connected at a
     distance) — a statistic */, MP_CTR_RESPAWNS, MP_CTR_BAD_ACTIONS, MP_CTR_COUNT
};
int mp_counters(MpEngine* eng, uint64_t out[MP_CTR_COUNT]);

/* Blocks until all work submitted on the engine's stream has finished. */
int mp_sync(MpEngine* eng);

/* Device memory for a view the caller is going to bind (unbind it before freeing
 * it).  chunk_bytes == 0: one hipMalloc.  chunk_bytes > 0: one virtual range mapped
 * onto separately created physical chunks of that size (HIP's virtual-memory API) —
 * another placement of the same bytes, and the speed of every step depends on where
 * the bound view lies (profiles/r04_write_fronts.md: the same launch, 99 - 122 us; a
 * property of the buffer's physical pages that no write order of the engine's removes).
 * (No reference counterpart: dmlab2d returns host arrays.) */
int mp_alloc_output(int device, uint64_t bytes, uint64_t chunk_bytes, void** out);
/* (views mapped from separately created 2 MB chunks are what the engine allocates for itself:
 * a physically contiguous view is written 25 - 45 % slower by the frame launch,
 * profiles/r05_alloc_method.md) */
/* (a mapped view's physical memory is released; its virtual range stays reserved for
 * the life of the process: reused ranges were seen to keep stale translations.  The
 * retired total is MpInfo.retired_va_bytes; mp_alloc_output / mp_place_output refuse to
 * map more once it would pass MpInfo.retired_va_limit — 16 TiB unless
 * mp_set_retired_va_limit says otherwise — with MP_ERR_HIP and a message that says so) */
int mp_free_output(int device, void* ptr);
int mp_set_retired_va_limit(int64_t bytes);
/* The two callbacks torch.cuda.memory.CUDAPluggableAllocator wants (signatures are
 * torch's): memory for a CALLER's tensors from the same scattered 2 MB chunks the engine
 * maps its own views from (requests of 32 MB and more; smaller ones are plain hipMalloc) —
 * a learner that must own the buffer it binds gets the engine's placement without handing
 * the allocation over.  meltingpot_amd/memory.py wraps them in a torch MemPool. */
void* mp_torch_alloc(long size, int device, void* stream);
void mp_torch_free(void* ptr, long size, int device, void* stream);

/* The plan follows the buffer.  Times the engine's candidate launch plans on the
 * pixel views bound right now and keeps the fastest for them (synchronises).  An
 * engine nothing has been done with yet (no reset, step or restore: the usual moment
 * to bind) is really stepped for it — all worlds reset, a few steps of uniformly random
 * actions per plan —
 * behind a device-side copy of the records, the counters and the engine's scalar outputs
 * that is put back on every path out of the call (the caller's bound scalar outputs are
 * unbound for the duration: nothing of the caller's but the pixel views being timed is
 * written); without room for that copy the probe runs dry; an
 * engine in use is timed dry (every bound view drawn exactly as a step draws it, no
 * world stepped, no record or scalar output written); a plan replaces the stock one
 * only by a margin (3 % stepped, 6 % dry).  Results never depend on the plan (ring depth,
 * worlds per batch, pooled share, store policy: frame.hip plan_frame); on an output
 * buffer the memory side serves unevenly a pooled plan is 3 - 8 % faster (sc1 stores
 * 13 % for commons_harvest), on an even one they are slower.  `us_per_launch` (may be NULL): the kept plan's time.  A no-op without a
 * bound pixel view, and for an engine created with MpConfig.dev (explicit plans).
 * With pixel views bound as a ring every slot is timed (it is its own buffer) and keeps its
 * own plan (us_per_launch: their mean) — the timed launches DRAW into the slots, so tune a
 * ring before the rollout it is going to hold.  Work in flight finishes first (a tune
 * between two steps is legal and leaves the records as they were). */
int mp_tune(MpEngine* eng, double* us_per_launch);

/* What mp_place_output measured. */
typedef struct {
  int32_t candidates;   /* buffers tried */
  int32_t picked;       /* index of the one kept */
  float us[32];         /* time per launch of each under the plan that suits it, us */
  int32_t stepped;      /* 1: timed with real steps behind a copy of the state (an engine
                           nothing had been done with AND room for the copy), 0: dry launches */
  /* (ABI 6) */
  int32_t requested;    /* candidates asked for (clamped to 1 .. 32, and to what max_bytes allows) */
  int32_t out_of_memory; /* > 0: a candidate could not be mapped (device memory, or the bound on
                           retired address space) and the probe went on with the ones it had */
  int32_t early_exit;   /* why fewer than `requested` were tried: 1 = a round (the first: eight) of candidates
                           all within 3 % (no lottery to win for this view on this box), 2 = a
                           candidate 8 % below the median was found, 0 = neither */
  float setup_ms;       /* wall time of the whole call: set-up cost a caller pays once */
} MpPlacement;

/* Allocates the output buffer of `kind` where this engine writes it fastest, and
 * binds it.  Up to `candidates` (<= 32) buffers of mp_obs_bytes(kind), each mapped
 * from 2 MB physical chunks — another scatter of pages each —, at most `max_bytes`
 * of them alive at any time (0: a quarter of the device's free memory; at least two
 * candidates are compared if memory allows), in rounds of up to twelve — a further
 * round only while no candidate stands out (8 % below the median); each is bound,
 * tuned (mp_tune) and timed; the fastest stays bound and is returned in *device_ptr,
 * the others are released before the call returns.  The caller frees the result with mp_free_output
 * after unbinding it (mp_bind_output(kind, NULL)) or destroying the engine.
 * A caller that binds its OWN buffer gets that buffer's speed; mp_tune is what it
 * can still do.  Synchronises.  Whatever happens — an error half way included — the engine's
 * records, counters and scalar outputs (its own and the caller's bound ones) are what they
 * were before the call, every candidate but the one returned is released, and on an error the
 * kind is bound to what it was bound to.  MP_ERR_INVALID when max_bytes does not hold one
 * view; when memory runs out half way the probe goes on with what it has and says so in
 * MpPlacement.out_of_memory. */
int mp_place_output(MpEngine* eng, MpObsKind kind, int32_t candidates, uint64_t max_bytes,
                    void** device_ptr, MpPlacement* report);

/* What the box's memory system gives the pixel view `kind` ON THE BUFFER BOUND RIGHT NOW
 * (a calibration for benchmark lines: the frame launch is HBM-write bound, boxes and buffers
 * differ — profiles/r05_alloc_method.md — and a line must be able to tell a slow box from a
 * regression).  Times, with events on the engine's stream, `reps` launches each of
 *   memset_us         hipMemsetAsync over the view (the runtime's own fill kernel);
 *   product_order_us  a bare store loop in the frame launch's write order: the current plan's
 *                     workgroups, each its own contiguous range of the view, its renderer
 *                     waves taking whole pass-sized spans (10 - 12 KB) from an LDS counter,
 *                     16-byte lane-contiguous non-temporal stores — no step, no drawing;
 *   front_4k_us       the same bytes as ONE chip-wide front of 4 KiB spans, one span per
 *                     workgroup and turn (the placement-insensitive order of
 *                     profiles/r05_kib_front.md).
 * OVERWRITES the view with junk (the next step or mp_observe redraws it); touches nothing
 * else of the engine.  Synchronises.  MP_ERR_INVALID unless `kind` is a pixel view bound with
 * mp_bind_output / mp_place_output (not a ring). */
typedef struct {
  uint64_t bytes;          /* the view's bytes = what each launch writes */
  float memset_us, product_order_us, front_4k_us;
  int32_t groups, waves;   /* the store loops' geometry: workgroups, storing waves each */
  uint32_t span_bytes;     /* product order: bytes per span (one renderer pass) */
} MpBoxFill;
int mp_box_fill(MpEngine* eng, MpObsKind kind, int32_t reps, MpBoxFill* out);

/* Diagnostics.  The frame kernel bounds every wait of its pipeline (2 s of wall
 * time); a wave that gives up records where in words 0-5 ({site, workgroup, wave,
 * batch, seen, wanted}; word 0 == 0: no stall), and every synchronising call above
 * reports it as MP_ERR_HIP until mp_reset(eng, seeds, NULL) — a reset of ALL
 * worlds — clears it (the worlds of a stalled launch are incomplete; resetting
 * them makes the engine usable again).  Word 8: 1 + the world that paid an
 * interaction reward outside every colour interval (the_matrix; reported once).  The words live in host memory: this call never touches the
 * device, so it answers even while a kernel is stuck.  (Words 16.. are used by
 * the -DMP_FRAME_TRACE developer build.) */
int mp_fault_words(const MpEngine* eng, uint32_t out[64]);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* MP_ENGINE_H_ */
