/* ORACLE — test infrastructure only.  See engine.h.
 *
 * C entry points loaded with ctypes by oracle/oracle.py.  Episode lifecycle
 * follows lua/modules/api_factory.lua:85-111 (api:start / api:advance).  The
 * reference rebuilds the environment with seed + 1 on every reset
 * (utils/substrates/builder.py:177-181, reset_wrapper.py:37-45); here every
 * episode has its own stream of the counter-based generator: key = world seed,
 * episode index in the counter (A10).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mp_pack.h"
#include "engine.h"
#include "mt19937_64.h"

static const int32_t* tab_i32(const void* pack, const char* name) {
  uint64_t n;
  return (const int32_t*)mpk_find(pack, name, &n, 0);
}

static void bare_noop(Oracle* o) { (void)o; }
static const SubstrateVtbl kBareVtbl = {0, 0, 0, bare_noop, bare_noop, bare_noop};

/* num_players: 0 = as many as the pack was lowered for, else the first
 * num_players avatars of the pack play (num_players = len(roles),
 * configs/substrates/clean_up.py:847). */
Oracle* orc_create_players(const void* pack, uint64_t len, uint64_t world_seed,
                           int num_players) {
  if (mpk_validate(pack, len) != 0) return 0;
  Oracle* o = (Oracle*)calloc(1, sizeof(Oracle));
  void* copy = malloc(len);
  memcpy(copy, pack, len);
  o->pack = copy;
  o->hdr = tab_i32(copy, "hdr");
  o->H = o->hdr[MPK_HDR_H]; o->W = o->hdr[MPK_HDR_W]; o->L = o->hdr[MPK_HDR_L];
  o->P_pack = o->hdr[MPK_HDR_P];
  o->P = num_players > 0 && num_players <= o->P_pack ? num_players
         : o->hdr[MPK_HDR_DEFAULT_P] > 0 && o->hdr[MPK_HDR_DEFAULT_P] <= o->P_pack
             ? o->hdr[MPK_HDR_DEFAULT_P] : o->P_pack;
  o->nstates = o->hdr[MPK_HDR_NSTATES];
  o->nsprites = o->hdr[MPK_HDR_NSPRITES];
  o->topology = o->hdr[MPK_HDR_TOPOLOGY];
  o->max_frames = o->hdr[MPK_HDR_MAXFRAMES];
  o->nobj = o->hdr[MPK_HDR_NOBJ]; o->nhits = o->hdr[MPK_HDR_NHITS];
  o->avatar_layer = o->hdr[MPK_HDR_AVATAR_LAYER];
  o->state_layer = tab_i32(copy, "state_layer");
  o->state_sprite = tab_i32(copy, "state_sprite");
  o->state_contact = tab_i32(copy, "state_contact");
  o->state_groups = (const uint32_t*)tab_i32(copy, "state_groups");
  o->sprite_rgba = (const uint8_t*)tab_i32(copy, "sprite_rgba");
  o->sprite_flags = tab_i32(copy, "sprite_flags");
  o->objects = tab_i32(copy, "objects");
  o->alive_state = tab_i32(copy, "avatar_alive_state");
  o->wait_state = tab_i32(copy, "avatar_wait_state");
  o->view_sprite_map = tab_i32(copy, "view_sprite_map");
  o->hit_state = tab_i32(copy, "hit_state");
  o->action_table = tab_i32(copy, "action_table");
  o->hit_state_dir = tab_i32(copy, "hit_state_dir");
  o->state_orient = tab_i32(copy, "state_orient");
  o->init_grid = (const uint8_t*)tab_i32(copy, "init_grid");
  /* group id of 'spawnPoints' */
  {
    uint64_t n;
    const char* names = (const char*)mpk_find(copy, "group_names", &n, 0);
    int g = 0;
    o->spawn_group_mask = 0;
    for (uint64_t i = 0; i < n;) {
      if (strcmp(names + i, "spawnPoints") == 0) o->spawn_group_mask = 1 << g;
      i += strlen(names + i) + 1;
      ++g;
    }
  }
  size_t cells = (size_t)o->L * o->H * o->W;
  o->pieces = (Piece*)calloc((size_t)o->nobj + 1, sizeof(Piece));
  o->cell = (int*)malloc(cells * sizeof(int));
  o->beam = (uint8_t*)calloc(cells, 1);
  o->world_seed = world_seed;
  o->episode = 0;
  o->opt_blocked_move_reenters = 1;
  o->opt_beam_marks_blocked = 1;
  o->opt_dead_view_black = 1;
  o->opt_shuffle_order = 1;
  o->opt_flush_count = ORC_FLUSH_COUNT;
  o->opt_teleport_free_only = 0;
  switch (o->hdr[MPK_HDR_SUBSTRATE]) {
    case MPK_SUBSTRATE_CLEAN_UP:
      o->sub = &kCleanUpVtbl;
      o->sub_state = clean_up_create(o);
      break;
    case MPK_SUBSTRATE_TERRITORY:
      o->sub = &kTerritoryVtbl;
      o->sub_state = territory_create(o);
      break;
    case MPK_SUBSTRATE_COMMONS_HARVEST:
      o->sub = &kCommonsVtbl;
      o->sub_state = commons_create(o);
      break;
    case MPK_SUBSTRATE_COINS:
      o->sub = &kCoinsVtbl;
      o->sub_state = coins_create(o);
      break;
    case MPK_SUBSTRATE_THE_MATRIX:
      o->sub = &kMatrixVtbl;
      o->sub_state = matrix_create(o);
      break;
    case MPK_SUBSTRATE_COOP_MINING:
      o->sub = &kCoopVtbl;
      o->sub_state = coop_create(o);
      break;
    case MPK_SUBSTRATE_GIFT_REFINEMENTS:
      o->sub = &kGiftVtbl;
      o->sub_state = gift_create(o);
      break;
    case MPK_SUBSTRATE_COLLABORATIVE_COOKING:
      o->sub = &kCookVtbl;
      o->sub_state = cook_create(o);
      break;
    case MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS:
      o->sub = &kMushroomVtbl;
      o->sub_state = mushroom_create(o);
      break;
    case 0: /* bare engine, no substrate rules: the reference's Lua KATs */
      o->sub = &kBareVtbl;
      break;
    default:
      free(o);
      return 0;
  }
  return o;
}

Oracle* orc_create(const void* pack, uint64_t len, uint64_t world_seed) {
  return orc_create_players(pack, len, world_seed, 0);
}

void orc_destroy(Oracle* o) {
  if (!o) return;
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_CLEAN_UP)
    clean_up_destroy(o->sub_state);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COMMONS_HARVEST)
    commons_destroy(o->sub_state);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_TERRITORY)
    territory_destroy(o->sub_state);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COINS)
    coins_destroy(o->sub_state);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_THE_MATRIX)
    matrix_destroy(o->sub_state);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COOP_MINING)
    coop_destroy(o->sub_state);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_GIFT_REFINEMENTS)
    gift_destroy(o->sub_state);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COLLABORATIVE_COOKING)
    cook_destroy(o->sub_state);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS)
    mushroom_destroy(o->sub_state);
  free(o->pieces); free(o->cell); free(o->beam); free((void*)o->pack); free(o->mt);
  free(o);
}

void orc_set_option(Oracle* o, int which, int value) {
  if (which == 0) o->opt_blocked_move_reenters = value;
  if (which == 1) o->opt_beam_marks_blocked = value;
  if (which == 2) o->opt_dead_view_black = value;
  if (which == 3) o->opt_shuffle_order = value;
  if (which == 4) o->opt_flush_count = value > 0 ? value : 1;
  if (which == 5) o->opt_teleport_free_only = value;
  if (which == 6) o->opt_serial_rng = value;
  if (which == 7) o->opt_serial_int_method = value;
  if (which == 8) o->opt_serial_shuffle_back = value;
}

/* the generator alone, for its known-answer test: the n-th output after seed */
uint64_t orc_mt19937_64(uint64_t seed, int n) {
  Mt64 g;
  mt64_seed(&g, seed);
  uint64_t x = 0;
  for (int i = 0; i < n; ++i) x = mt64_next(&g);
  return x;
}
/* its conversions: kind 0 = uniformReal as 53 bits, 1 / 2 = uniformInt(0, n - 1) by method 0 / 1 */
uint64_t orc_mt19937_64_draw(uint64_t seed, int skip, int kind, uint64_t n) {
  Mt64 g;
  mt64_seed(&g, seed);
  for (int i = 0; i < skip; ++i) (void)mt64_next(&g);
  return kind == 0 ? mt64_u53(&g) : mt64_bounded(&g, n, kind - 1);
}

/* api:start(episode, seed) (api_factory.lua:85-102) +
 * BaseSimulation:start/_avatarStart (base_simulation.lua:396-471). */
void orc_reset(Oracle* o) {
  o->ep = o->episode++;
  o->k0 = (uint32_t)o->world_seed; o->k1 = (uint32_t)(o->world_seed >> 32);
  if (o->opt_serial_rng) eng_reseed(o);   /* random:seed(seed), api_factory.lua:89 */
  o->frame = 0; o->step = 0; o->continue_flag = 1; o->done = 0;
  memset(o->num_zapped, 0, sizeof o->num_zapped);
  memset(o->zap_matrix, 0, sizeof o->zap_matrix);
  o->ev_count = 0;
  o->qlen[0] = o->qlen[1] = 0; o->qcur = 0;
  o->npieces = 0;
  size_t cells = (size_t)o->L * o->H * o->W;
  for (size_t i = 0; i < cells; ++i) o->cell[i] = -1;
  memset(o->beam, 0, cells);

  /* start() on all non-avatar objects in creation order (scene first).  Objects
   * of a 'choice' map character (prefab_utils.lua:101-103: random:choice(list)
   * once per world build, i.e. per episode) exist only in the outcomes their
   * mask lists; outcome of choice c = draw (RS_MAP_CHOICE, index c), bounded by
   * the list length.  The per-kind index counts absent objects too, so it stays
   * the index into the pack's per-kind tables. */
  int counters[32] = {0};
  uint64_t n_choice = 0;
  const int32_t* choice_n = (const int32_t*)mpk_find(o->pack, "choice_n", &n_choice, 0);
  const int32_t* obj_choice = (const int32_t*)mpk_find(o->pack, "object_choice", 0, 0);
  const int32_t* obj_choice_hi = (const int32_t*)mpk_find(o->pack, "object_choice_hi", 0, 0);
  for (int i = 0; i < o->nobj; ++i) {
    const int32_t* ob = o->objects + 4 * i;
    int kind = ob[0];
    if (kind == MPK_KIND_AVATAR) continue;
    int idx = counters[kind & 31]++;
    if (choice_n && obj_choice[2 * i] >= 0) {
      int cid = obj_choice[2 * i];
      /* choice_n < 0: a choice the config makes when it BUILDS the environment
       * (coins.py:45-82,500: map size drawn in build()): one outcome per world, the
       * same in all its episodes — the draw does not carry the episode */
      int n = choice_n[cid];
      PhiloxOut d = eng_draw(o, RS_MAP_CHOICE, (uint32_t)cid);
      if (n < 0) {
        n = -n;
        d = philox4x32_10((uint32_t)cid, RS_MAP_CHOICE, 0u, 0xffffffffu, o->k0, o->k1);
      }
      int k = (int)eng_bounded(o, d, (uint32_t)n);
      uint64_t mask = (uint32_t)obj_choice[2 * i + 1];
      if (obj_choice_hi) mask |= (uint64_t)(uint32_t)obj_choice_hi[i] << 32;
      if (!((mask >> k) & 1)) continue;
    }
    eng_create_piece(o, ob[3], ob[1], ob[2], ORIENT_N, kind, idx);
  }
  /* _avatarStart (base_simulation.lua:396-445): for every initial spawn group
   * groupShuffledWithCount(random, group, #avatars of the group) — a partial
   * Fisher-Yates over the group's pieces in creation order — and avatar i takes
   * the next sampled point of its group (the reference iterates groups and
   * avatars with pairs(): unspecified order, fixed here to first use / player
   * index, Appendix B).  Draw index = position + 256 * group. */
  int spawn_cell[ORC_MAX_PLAYERS];
  {
    uint64_t ncells, nptr;
    const int32_t* cells = (const int32_t*)mpk_find(o->pack, "init_spawn_cells", &ncells, 0);
    const int32_t* ptr = (const int32_t*)mpk_find(o->pack, "init_spawn_ptr", &nptr, 0);
    const int32_t* grp = (const int32_t*)mpk_find(o->pack, "avatar_init_group", 0, 0);
    const uint32_t* gmask = (const uint32_t*)mpk_find(o->pack, "init_spawn_mask", 0, 0);
    for (int g = 0; g + 1 < (int)nptr; ++g) {
      int pool[1024], ns = 0, want = 0, taken = 0;
      for (int i = ptr[g]; i < ptr[g + 1]; ++i) {
        /* (a spawn point of a 'choice' character may not exist this episode: it
         * does iff a piece of the spawn group stands on its cell) */
        int present = choice_n == 0;
        for (int l = 0; l < o->L && !present; ++l) {
          int q = o->cell[((size_t)l * o->H + cells[i] / o->W) * o->W + cells[i] % o->W];
          present = q >= 0 && (o->state_groups[o->pieces[q].state] & gmask[g]) != 0;
        }
        if (present) pool[ns++] = cells[i];
      }
      for (int p = 0; p < o->P; ++p) want += grp[p] == g;
      if (ns < want) abort(); /* "Insufficient spawn points!" */
      for (int i = 0; i < want; ++i) {
        int j = i + (int)eng_bounded(o, 
            eng_draw(o, RS_START_SPAWN, (uint32_t)(i + 256 * g)), (uint32_t)(ns - i));
        int t = pool[i]; pool[i] = pool[j]; pool[j] = t;
      }
      for (int p = 0; p < o->P; ++p)
        if (grp[p] == g) spawn_cell[p] = pool[taken++];
    }
  }
  for (int p = 0; p < o->P; ++p) {
    /* Avatar:start (avatar_library.lua:288-320): random:choice(_COMPASS) */
    int orient = (int)eng_bounded(o, eng_draw(o, RS_START_ORIENT, (uint32_t)p), 4u);
    o->avatar_piece[p] = eng_create_piece(o, o->alive_state[p],
                                          spawn_cell[p] % o->W, spawn_cell[p] / o->W,
                                          orient, MPK_KIND_AVATAR, p);
    o->reward[p] = 0.0;
    o->movement_allowed[p] = 1;
    o->freeze_counter[p] = o->removal_counter[p] = 0;
    o->zap_timer[p] = 0; /* Zapper:start (avatar_library.lua:698-707) */
    eng_event(o, 9 /* AvatarStarted, avatar_library.lua:317 */, 0, 0);
    for (int a = 0; a < 4; ++a) o->action[p][a] = 0; /* action defaults */
  }
  o->sub->start(o);
  eng_do_update(o); /* api_factory.lua:101 */
}

static int advance(Oracle* o) {
  o->sub->sim_update(o);
  eng_do_update(o);
  int cont = o->continue_flag && o->step < o->max_frames;
  o->done = !cont;
  return cont;
}

static void begin_advance(Oracle* o) {
  o->ev_count = 0;
  o->step++;
  memset(o->num_zapped, 0, sizeof o->num_zapped);   /* Zapper:update */
  memset(o->zap_matrix, 0, sizeof o->zap_matrix);   /* GlobalMetricHolder:update */
}

/* A step asked of a world whose episode has ended and that is not reset (the reference has no
 * such call: dm_env restarts on the step after LAST; include/mp_engine.h MpConfig.auto_reset = 0
 * keeps the world as it is "until mp_reset"): nothing moves, and the step reports no reward and
 * no events — what stepk::dispatch does for a frozen world (csrc/step_common.h). */
static int frozen_step(Oracle* o) {
  for (int p = 0; p < o->P; ++p) o->reward[p] = 0.0;
  o->ev_count = 0;
  return 0;
}

/* api:discreteActions + api:advance (api_factory.lua:81-111).  `actions` are
 * discrete ids into ACTION_SET (discrete_action_wrapper.py:97-109).  Returns
 * the continue flag. */
int orc_step(Oracle* o, const int32_t* actions) {
  if (o->done) return frozen_step(o);
  begin_advance(o);
  for (int p = 0; p < o->P; ++p)
    for (int a = 0; a < 4; ++a)
      o->action[p][a] = o->action_table[actions[p] * 4 + a];
  return advance(o);
}

/* The same with the raw fields dmlab2d hands to Avatar:discreteActions
 * (avatar_library.lua:217-223): `fields` is [P][nfields] in actionOrder.  A
 * player with a field outside its actionSpec range (table "action_spec": min,
 * max, default per field) acts with the defaults — dmlab2d would refuse the
 * value at its Python boundary; include/mp_engine.h mp_step_fields. */
int orc_step_fields(Oracle* o, const int32_t* fields) {
  if (o->done) return frozen_step(o);
  const int32_t* spec = tab_i32(o->pack, "action_spec");
  const int nf = o->hdr[MPK_HDR_NFIELDS];
  begin_advance(o);
  for (int p = 0; p < o->P; ++p) {
    int ok = 1;
    for (int a = 0; a < nf; ++a)
      ok = ok && fields[p * nf + a] >= spec[3 * a] && fields[p * nf + a] <= spec[3 * a + 1];
    for (int a = 0; a < 4; ++a)
      o->action[p][a] = a < nf ? (ok ? fields[p * nf + a] : spec[3 * a + 2]) : 0;
  }
  return advance(o);
}

int orc_done(const Oracle* o) { return o->done; }

/* Trace fitting (tests/tools/replay_trace.py): puts avatar `p` where a recorded
 * trajectory has it — on the grid at (x, y) facing `orient`, or off the grid
 * (alive == 0) — without firing callbacks.  Returns 0 if the cell is taken. */
int orc_place_avatar(Oracle* o, int p, int x, int y, int orient, int alive) {
  int piece = o->avatar_piece[p];
  Piece* pc = &o->pieces[piece];
  int layer = o->state_layer[pc->state];
  const int old_x = pc->x, old_y = pc->y;
  if (alive) {   /* (checked before anything moves: a refused placement changes nothing) */
    const int al = o->state_layer[o->alive_state[p]];
    if (x < 0 || x >= o->W || y < 0 || y >= o->H) return 0;
    const int there = o->cell[((size_t)al * o->H + y) * o->W + x];
    if (there >= 0 && there != piece) return 0;
  }
  if (layer >= 0) o->cell[((size_t)layer * o->H + pc->y) * o->W + pc->x] = -1;
  pc->orient = orient & 3;
  if (!alive) {
    if (pc->state != o->wait_state[p]) { pc->state = o->wait_state[p]; pc->change_frame = o->frame; }
    return 1;
  }
  if (pc->state != o->alive_state[p]) { pc->state = o->alive_state[p]; pc->change_frame = o->frame; }
  layer = o->state_layer[pc->state];
  size_t ci = ((size_t)layer * o->H + y) * o->W + x;
  pc->x = x; pc->y = y;
  o->cell[ci] = piece;
  /* the pieces connected to it (grid:connect) come along and face the same way */
  for (int q = 0; q < o->npieces; ++q) {
    Piece* f = &o->pieces[q];
    if (f->leader != piece) continue;
    const int fl = o->state_layer[f->state];
    if (fl >= 0 && f->x == old_x && f->y == old_y) {
      o->cell[((size_t)fl * o->H + f->y) * o->W + f->x] = -1;
      o->cell[((size_t)fl * o->H + y) * o->W + x] = q;
    }
    f->x = x; f->y = y; f->orient = orient & 3;
  }
  return 1;
}
/* (test hook) the piece at (layer, x, y) put in `state` — a state of the same layer — at once,
 * behind the engine's back: scripted situations (an ore of a given kind next to two avatars)
 * that a rollout would take thousands of frames to reach.  Returns 0 if there is no piece. */
int orc_set_cell_state(Oracle* o, int layer, int x, int y, int state) {
  if (layer < 0 || layer >= o->L || x < 0 || x >= o->W || y < 0 || y >= o->H) return 0;
  int piece = o->cell[((size_t)layer * o->H + y) * o->W + x];
  if (piece < 0 || state <= 0 || state >= o->nstates || o->state_layer[state] != layer) return 0;
  if (o->pieces[piece].state != state) { o->pieces[piece].state = state; o->pieces[piece].change_frame = o->frame; }
  return 1;
}
/* api:events of the last reset / advance: up to `cap` rows {type, a, b}; returns
 * the number of events that were added (may exceed cap). */
int orc_events(const Oracle* o, int32_t* out, int cap) {
  int n = o->ev_count < ORC_MAX_EVENTS ? o->ev_count : ORC_MAX_EVENTS;
  for (int i = 0; i < n && i < cap; ++i)
    for (int k = 0; k < 3; ++k) out[3 * i + k] = o->ev[i][k];
  return o->ev_count;
}
int orc_step_count(const Oracle* o) { return o->step; }
/* "priority:tag\n" per updater the last grid:update ran, in order; returns the
 * length written (truncated to cap - 1). */
int orc_updater_trace(const Oracle* o, char* buf, int cap) {
  int n = 0;
  for (int i = 0; i < o->trace_n && n < cap - 1; ++i)
    n += snprintf(buf + n, (size_t)(cap - n), "%d:%s\n", o->trace[i].priority, o->trace[i].tag);
  return n < cap ? n : cap - 1;
}
void orc_rewards(const Oracle* o, double* out) {
  for (int p = 0; p < o->P; ++p) out[p] = o->reward[p];
}

/* Zapper:readyToShoot (avatar_library.lua:737-744) */
void orc_ready_to_shoot(const Oracle* o, double* out) {
  uint64_t n;
  const int32_t* zi = (const int32_t*)mpk_find(o->pack, "zapper_i32", &n, 0);
  for (int p = 0; p < o->P; ++p) {
    if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_THE_MATRIX) {
      out[p] = matrix_ready_to_shoot(o, p);
      continue;
    }
    int alive = o->pieces[o->avatar_piece[p]].state == o->alive_state[p];
    /* (a level without a Zapper has no such observation: timer 0, cooldown 1; coop_mining's
     * ReadyToShootObservation reads its MineBeam, components.lua:172-175) */
    const int cooldown = o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COOP_MINING ? coop_cooldown(o)
                         : o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_GIFT_REFINEMENTS ? gift_cooldown(o)
                         : o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COLLABORATIVE_COOKING ? cook_cooldown(o)
                         : zi ? zi[0] : 1;
    double v = 1.0 - (double)o->zap_timer[p] / (double)cooldown;
    out[p] = alive ? (v > 0.0 ? v : 0.0) : 0.0;
  }
}

/* clean_up's debug metrics (clean_up.py:751-784): out[4][P] = PLAYER_CLEANED,
 * PLAYER_ATE_APPLE, NUM_OTHERS_PLAYER_ZAPPED_THIS_STEP, NUM_OTHERS_WHO_ATE_THIS_STEP */
void orc_debug_metrics(const Oracle* o, double* out) {
  for (int p = 0; p < o->P; ++p) {
    int cu = o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_CLEAN_UP;
    out[0 * o->P + p] = cu ? clean_up_debug_metric(o, p, 0) : 0.0;
    out[1 * o->P + p] = cu ? clean_up_debug_metric(o, p, 1) : 0.0;
    out[2 * o->P + p] = (double)o->num_zapped[p];
    out[3 * o->P + p] = cu ? clean_up_debug_metric(o, p, 2) : 0.0;
  }
}
/* playerZapMatrix(zapped, zapper) of the step: out[P][P] */
void orc_zap_matrix(const Oracle* o, double* out) {
  for (int v = 0; v < o->P; ++v)
    for (int z = 0; z < o->P; ++z) out[v * o->P + z] = (double)o->zap_matrix[v][z];
}

void orc_num_others_cleaned(const Oracle* o, double* out) {
  for (int p = 0; p < o->P; ++p)
    out[p] = o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_CLEAN_UP
                 ? clean_up_num_others_cleaned(o, p)
             : o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COINS
                 ? coins_partner_mismatch(o, p) : 0.0;
}

/* Canonical state dump compared bit-for-bit against the engine's:
 *   grid  u8[L][H][W]  state id of the piece (or beam pseudo-state) per cell
 *   avat  i32[P][8]    x, y, orient, alive, zap_timer, aux_timer,
 *                      frames_in_state, 0
 *   glob  i32[8]       step, done, frame, dirt_count, episode, 0, 0, 0   */
void orc_dump(const Oracle* o, uint8_t* grid, int32_t* avat, int32_t* glob) {
  size_t cells = (size_t)o->L * o->H * o->W;
  for (size_t i = 0; i < cells; ++i) {
    int piece = o->cell[i];
    grid[i] = piece >= 0 ? (uint8_t)o->pieces[piece].state : o->beam[i];
  }
  for (int p = 0; p < o->P; ++p) {
    const Piece* pc = &o->pieces[o->avatar_piece[p]];
    int32_t* a = avat + 8 * p;
    a[0] = pc->x; a[1] = pc->y; a[2] = pc->orient;
    a[3] = pc->state == o->alive_state[p];
    a[4] = o->zap_timer[p];
    a[5] = o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_CLEAN_UP
               ? clean_up_clean_timer(o, p)
           : o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_TERRITORY
               ? territory_claim_timer(o, p) : 0;
    a[6] = eng_frames(o, o->avatar_piece[p]);
    a[7] = 0;
  }
  glob[0] = o->step; glob[1] = o->done; glob[2] = o->frame;
  glob[3] = o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_CLEAN_UP
                ? clean_up_dirt_count(o)
            : o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COMMONS_HARVEST
                ? commons_live_apples(o)
            : o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COINS ? coins_live(o) : 0;
  glob[4] = (int32_t)o->episode; glob[5] = glob[6] = glob[7] = 0;
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_TERRITORY) territory_dump(o, avat, glob);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_THE_MATRIX) matrix_dump(o, avat, glob);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COOP_MINING) coop_dump(o, glob);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_GIFT_REFINEMENTS) gift_dump(o, avat, glob);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_COLLABORATIVE_COOKING) cook_dump(o, grid, glob);
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_EXTERNALITY_MUSHROOMS) mushroom_dump(o, avat, glob);
}

/* *_in_the_matrix observations: "N.INVENTORY" f64 [P][R] and
 * "N.INTERACTION_INVENTORIES" f64 [P][2][R]; returns R (0 for other levels) */
int orc_inventories(const Oracle* o, double* inventory, double* interaction) {
  if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_GIFT_REFINEMENTS) {
    /* gift_refinements: "N.INVENTORY" only (AvatarMetricReporter on Inventory.inventory) */
    const int K = gift_num_types(o);
    for (int p = 0; p < o->P; ++p) gift_inventory(o, p, inventory + (size_t)p * K);
    return K;
  }
  if (o->hdr[MPK_HDR_SUBSTRATE] != MPK_SUBSTRATE_THE_MATRIX) return 0;
  const int R = matrix_num_resources(o);
  for (int p = 0; p < o->P; ++p) {
    matrix_inventory(o, p, inventory + (size_t)p * R);
    matrix_interaction_inventories(o, p, interaction + (size_t)p * 2 * R);
  }
  return R;
}
/* (row_reward, col_reward) of each player's latest 'interaction' event: out[P][2] */
void orc_interaction_rewards(const Oracle* o, double* out) {
  for (int p = 0; p < o->P; ++p) {
    out[2 * p] = out[2 * p + 1] = 0.0;
    if (o->hdr[MPK_HDR_SUBSTRATE] == MPK_SUBSTRATE_THE_MATRIX) matrix_interaction_rewards(o, p, out + 2 * p);
  }
}
/* the cumulants of the_matrix.get_cumulant_metric_configs: out[P][1 + 3 R] */
void orc_matrix_cumulants(const Oracle* o, double* out) {
  const int R = matrix_num_resources(o), n = 1 + 3 * R;
  for (int p = 0; p < o->P; ++p)
    for (int k = 0; k < n; ++k) out[p * n + k] = matrix_cumulant(o, p, k);
}

void orc_render_agent(const Oracle* o, int player, uint8_t* rgb) {
  orc_render_view(o, player, rgb);
}
void orc_render_world_rgb(const Oracle* o, uint8_t* rgb) { orc_render_world(o, rgb); }

/* ---- direct engine access for the reference's Lua KATs
 * (tests/test_oracle_reference_kats.py) ---- */
int orc_piece_x(const Oracle* o, int piece) { return o->pieces[piece].x; }
int orc_piece_y(const Oracle* o, int piece) { return o->pieces[piece].y; }
int orc_piece_orient(const Oracle* o, int piece) { return o->pieces[piece].orient; }
int orc_piece_state(const Oracle* o, int piece) { return o->pieces[piece].state; }
int orc_avatar_piece(const Oracle* o, int p) { return o->avatar_piece[p]; }
void orc_q_move_abs(Oracle* o, int piece, int d) { eng_move_abs(o, piece, d); }
void orc_q_move_rel(Oracle* o, int piece, int d) { eng_move_rel(o, piece, d); }
void orc_q_turn(Oracle* o, int piece, int q) { eng_turn(o, piece, q); }
void orc_q_set_orientation(Oracle* o, int piece, int d) { eng_set_orientation(o, piece, d); }
void orc_q_teleport(Oracle* o, int piece, int x, int y) { eng_teleport(o, piece, x, y); }
void orc_q_set_state(Oracle* o, int piece, int s) { eng_set_state(o, piece, s); }
void orc_q_teleport_to_group(Oracle* o, int piece, uint32_t mask, int state,
                             int mode) {
  eng_teleport_to_group(o, piece, mask, state, mode, RS_RESPAWN, 0);
}
void orc_q_hit_beam(Oracle* o, int piece, int hit, int len, int rad) {
  eng_hit_beam(o, piece, hit, len, rad);
}
void orc_grid_update(Oracle* o) { eng_do_update(o); }
void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                uint32_t k1, uint32_t* out) {
  PhiloxOut r = philox4x32_10(c0, c1, c2, c3, k0, k1);
  memcpy(out, r.x, 16);
}
