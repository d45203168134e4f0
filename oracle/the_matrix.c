/* ORACLE — test infrastructure only.  See engine.h.
 *
 * *_in_the_matrix rules: restatement of the reference's Lua components
 *   lua/levels/the_matrix/components.lua  (Resource, Destroyable, TheMatrix,
 *     SpawnResourcesWhenAllPlayersZapped, GameInteractionZapper, Taste,
 *     InteractionTaste, DyadicRole, ReadyToInteractMarker)
 *   lua/modules/avatar_library.lua        (Avatar incl. the timed freeze,
 *     :884-945 AvatarConnector)
 *   lua/modules/component_library.lua:907-948 (StochasticIntervalEpisodeEnding)
 * with kwargs from configs/substrates/<game>_in_the_matrix__<variant>.py and
 * the_matrix.py (in the pack: mx_*).
 */
#include <stdlib.h>
#include <string.h>

#include "../include/mp_pack.h"
#include "engine.h"

enum { ACT_MOVE = 0, ACT_TURNA = 1, ACT_INTERACT = 2 };
enum { IND_NOT_READY = 0, IND_READY = 1, IND_COLOR1 = 2 }; /* TheMatrix.indicators */
#define MX_MAX_R 3

/* The effects one resolved interaction schedules on the ZAPPED player's
 * component (`self._scheduledEffects`, components.lua:634-693). */
typedef struct {
  int pending;
  int row, col, row_won;
  double row_reward, col_reward;
} Effects;

typedef struct {
  int R, n_site;
  int *site_piece, *site_class; /* per site (pack order); piece -1: not in this episode's map */
  int* health;                  /* Destroyable._variables.health */
  int mark_piece[ORC_MAX_PLAYERS];
  /* TheMatrix (components.lua:229-246) */
  double inv[ORC_MAX_PLAYERS][MX_MAX_R]; /* playerResources */
  int collected[ORC_MAX_PLAYERS];        /* playerCollectedAtLeastOneResource */
  int indicator[ORC_MAX_PLAYERS];        /* indicators */
  /* GameInteractionZapper (components.lua:873-892) */
  int interacted_flag[ORC_MAX_PLAYERS];  /* interactedThisStep (the blocker) */
  int till_effects[ORC_MAX_PLAYERS];     /* _framesTillScheduledEffects */
  double color_reward[ORC_MAX_PLAYERS];  /* _rewardToDetermineColor */
  int end_next_frame[ORC_MAX_PLAYERS];   /* _endEpisodeOnNextFrame */
  Effects fx[ORC_MAX_PLAYERS];
  double latest[ORC_MAX_PLAYERS][2][MX_MAX_R]; /* latest_interaction_inventories */
  double latest_rewards[ORC_MAX_PLAYERS][2];   /* (row_reward, col_reward) of a player's latest
                                                  interaction: the 'interaction' event's payload */
  /* cumulants (components.lua:840-853): interacted, collected_k, destroyed_k, argmax_k */
  int cum[ORC_MAX_PLAYERS][1 + 3 * MX_MAX_R];
  int ee_t;
  /* states */
  int s_mark_wait, s_ind[7], s_visible[MX_MAX_R], s_wait[MX_MAX_R];
  /* constants */
  int cooldown, beam_length, beam_radius, respawn_frames, freeze, end_on_first,
      reset_winner, reset_loser, loser_dies, winner_dies, zero_inventory, random_tie,
      disallow_unready, has_ee, ee_min_frames, ee_interval, regen_delay, initial_health,
      n_intervals, spawn_all, hit;
  double reward_floor, reward_multiplier, reward_unready;
  double row_matrix[MX_MAX_R][MX_MAX_R], col_matrix[MX_MAX_R][MX_MAX_R];
  double interval[5][2];
  uint64_t thr_regen, thr_ee;
  int taste_class[ORC_MAX_PLAYERS], itaste_class[ORC_MAX_PLAYERS],
      itaste_zero[ORC_MAX_PLAYERS], role[ORC_MAX_PLAYERS];
  double taste_reward[ORC_MAX_PLAYERS], taste_default[ORC_MAX_PLAYERS],
      itaste_extra[ORC_MAX_PLAYERS];
  const uint32_t* state_hit_block;
} Matrix;

static Matrix* mx(const Oracle* o) { return (Matrix*)o->sub_state; }

void* matrix_create(Oracle* o) {
  Matrix* c = (Matrix*)calloc(1, sizeof(Matrix));
  uint64_t n;
  const int32_t* st = (const int32_t*)mpk_find(o->pack, "mx_states", &n, 0);
  const int32_t* ci = (const int32_t*)mpk_find(o->pack, "mx_i32", &n, 0);
  const double* cf = (const double*)mpk_find(o->pack, "mx_f64", &n, 0);
  const uint64_t* thr = (const uint64_t*)mpk_find(o->pack, "mx_thr", &n, 0);
  const int32_t* pi = (const int32_t*)mpk_find(o->pack, "mx_player_i32", &n, 0);
  const double* pf = (const double*)mpk_find(o->pack, "mx_player_f64", &n, 0);
  const int R = ci[0];
  if (R < 1 || R > MX_MAX_R) abort();
  c->R = R;
  c->cooldown = ci[1]; c->beam_length = ci[2]; c->beam_radius = ci[3];
  c->respawn_frames = ci[4]; c->freeze = ci[5]; c->end_on_first = ci[6];
  c->reset_winner = ci[7]; c->reset_loser = ci[8]; c->loser_dies = ci[9];
  c->winner_dies = ci[10]; c->zero_inventory = ci[11]; c->random_tie = ci[12];
  c->disallow_unready = ci[13]; c->has_ee = ci[14]; c->ee_min_frames = ci[15];
  c->ee_interval = ci[16]; c->regen_delay = ci[17]; c->initial_health = ci[18];
  c->n_intervals = ci[19]; c->spawn_all = ci[20]; c->hit = ci[21];
  c->reward_floor = cf[0]; c->reward_multiplier = cf[1]; c->reward_unready = cf[2];
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < R; ++j) {
      c->row_matrix[i][j] = cf[5 + i * R + j];
      c->col_matrix[i][j] = cf[5 + R * R + i * R + j];
    }
  for (int k = 0; k < c->n_intervals; ++k) {
    c->interval[k][0] = cf[5 + 2 * R * R + 2 * k];
    c->interval[k][1] = cf[5 + 2 * R * R + 2 * k + 1];
  }
  c->thr_regen = thr[0]; c->thr_ee = thr[1];
  c->s_mark_wait = st[0];
  for (int k = 0; k < 7; ++k) c->s_ind[k] = st[1 + k]; /* ready, notReady, colour 1..5 */
  { int t = c->s_ind[0]; c->s_ind[0] = c->s_ind[1]; c->s_ind[1] = t; } /* -> IND_* order */
  for (int k = 0; k < R; ++k) { c->s_visible[k] = st[8 + 2 * k]; c->s_wait[k] = st[9 + 2 * k]; }
  for (int p = 0; p < o->P_pack && p < ORC_MAX_PLAYERS; ++p) {
    c->taste_class[p] = pi[4 * p]; c->itaste_class[p] = pi[4 * p + 1];
    c->itaste_zero[p] = pi[4 * p + 2]; c->role[p] = pi[4 * p + 3];
    c->taste_reward[p] = pf[4 * p]; c->taste_default[p] = pf[4 * p + 1];
    c->itaste_extra[p] = pf[4 * p + 2];
  }
  c->state_hit_block = (const uint32_t*)mpk_find(o->pack, "state_hit_block", &n, 0);
  const int32_t* cls = (const int32_t*)mpk_find(o->pack, "resource_class", &n, 0);
  c->n_site = (int)n;
  c->site_piece = (int*)malloc((size_t)c->n_site * sizeof(int));
  c->site_class = (int*)malloc((size_t)c->n_site * sizeof(int));
  c->health = (int*)malloc((size_t)c->n_site * sizeof(int));
  for (int i = 0; i < c->n_site; ++i) c->site_class[i] = cls[i] - 1;
  return c;
}

void matrix_destroy(void* s) {
  Matrix* c = (Matrix*)s;
  if (!c) return;
  free(c->site_piece); free(c->site_class); free(c->health); free(c);
}

/* "N.INVENTORY" (InventoryObserver, components.lua:942-963) */
void matrix_inventory(const Oracle* o, int p, double* out) {
  for (int k = 0; k < mx(o)->R; ++k) out[k] = mx(o)->inv[p][k];
}
/* "N.INTERACTION_INVENTORIES" (AvatarMetricReporter on
 * GameInteractionZapper.latest_interaction_inventories): [2][R], self first */
void matrix_interaction_inventories(const Oracle* o, int p, double* out) {
  const Matrix* c = mx(o);
  for (int s = 0; s < 2; ++s)
    for (int k = 0; k < c->R; ++k) out[s * c->R + k] = c->latest[p][s][k];
}
/* row_reward / col_reward of the 'interaction' event (reportEventAndCumulants,
 * components.lua:785-797) player p last took part in: [2] */
void matrix_interaction_rewards(const Oracle* o, int p, double* out) {
  out[0] = mx(o)->latest_rewards[p][0]; out[1] = mx(o)->latest_rewards[p][1];
}
/* READY_TO_SHOOT (GameInteractionZapper:readyToShoot, components.lua:914-917) */
double matrix_ready_to_shoot(const Oracle* o, int p) {
  return 1.0 - (double)o->zap_timer[p] / (double)mx(o)->cooldown;
}
/* the cumulants of the_matrix.get_cumulant_metric_configs: which = 0
 * INTERACTED_THIS_STEP, then per class k: 1 + 3k COLLECTED_RESOURCE_k+1,
 * 2 + 3k DESTROYED_RESOURCE_k+1, 3 + 3k ARGMAX_INTERACTION_INVENTORY_WAS_k+1 */
double matrix_cumulant(const Oracle* o, int p, int which) { return (double)mx(o)->cum[p][which]; }
int matrix_num_resources(const Oracle* o) { return mx(o)->R; }

/* Extra parity fields of the canonical dump (mirrored by mp_dump):
 *   avat[p][5] = till_effects + 1 | indicator << 8 | collected << 12 |
 *                movementAllowed << 13 | freeze << 16
 *   avat[p][7] = marker: on grid | x << 1 | y << 9 | state << 17
 *   glob[3] = live resources, glob[5] = sum of live resources' health */
void matrix_dump(const Oracle* o, int32_t* avat, int32_t* glob) {
  const Matrix* c = mx(o);
  for (int p = 0; p < o->P; ++p) {
    avat[8 * p + 5] = (c->till_effects[p] + 1) | (c->indicator[p] << 8) |
                      (c->collected[p] << 12) | (o->movement_allowed[p] << 13) |
                      (o->freeze_counter[p] << 16);
    const Piece* m = &o->pieces[c->mark_piece[p]];
    int on = o->state_layer[m->state] >= 0;
    avat[8 * p + 7] = on ? (1 | (m->x << 1) | (m->y << 9) | (m->state << 17)) : 0;
  }
  int live = 0, health = 0;
  for (int i = 0; i < c->n_site; ++i) {
    if (c->site_piece[i] < 0) continue;
    if (o->pieces[c->site_piece[i]].state == c->s_visible[c->site_class[i]]) {
      ++live; health += c->health[i];
    }
  }
  glob[3] = live; glob[5] = health;
}

static int is_alive(const Oracle* o, int p) {
  return o->pieces[o->avatar_piece[p]].state == o->alive_state[p];
}
static int is_wait(const Oracle* o, int p) {
  return o->pieces[o->avatar_piece[p]].state == o->wait_state[p];
}
/* Avatar:addReward with skipWaitStateRewards = false (avatar_library.lua:364-379;
 * every *_in_the_matrix config sets it): rewards reach avatars in the wait state */
static void add_reward(Oracle* o, int p, double amount) { o->reward[p] += amount; }

/* TheMatrix:resetInventory (components.lua:271-280) */
static void reset_inventory(Matrix* c, int p) {
  for (int k = 0; k < c->R; ++k) c->inv[p][k] = c->zero_inventory ? 0.0 : 1.0;
  c->collected[p] = 0;
}

static void reset_cumulants(Matrix* c, int p) { /* _resetBinaryCumulants (:840-853) */
  memset(c->cum[p], 0, sizeof c->cum[p]);
}

static void mx_start(Oracle* o) {
  Matrix* c = mx(o);
  for (int i = 0; i < c->n_site; ++i) { c->site_piece[i] = -1; c->health[i] = 0; }
  for (int i = 0; i < o->npieces; ++i) {
    int idx = o->pieces[i].index;
    if (o->pieces[i].kind == MPK_KIND_RESOURCE) {
      c->site_piece[idx] = i;
      c->health[idx] = c->initial_health; /* Destroyable:reset */
    } else if (o->pieces[i].kind == MPK_KIND_READY_MARKER) {
      c->mark_piece[idx] = i;
    }
  }
  c->ee_t = 1;
  for (int p = 0; p < o->P; ++p) {
    /* TheMatrix:reset (components.lua:229-246) */
    for (int k = 0; k < c->R; ++k) c->inv[p][k] = c->zero_inventory ? 0.0 : 1.0;
    c->collected[p] = 0;
    c->indicator[p] = IND_NOT_READY;
    /* GameInteractionZapper:start (:873-892) */
    o->zap_timer[p] = 0;
    for (int s = 0; s < 2; ++s)
      for (int k = 0; k < MX_MAX_R; ++k) c->latest[p][s][k] = 0.0; /* a fresh DoubleTensor */
    reset_cumulants(c, p);
    c->fx[p].pending = 0;
    c->till_effects[p] = -1;
    c->end_next_frame[p] = 0;
    c->interacted_flag[p] = 0;
    c->color_reward[p] = 0.0;
  }
  for (int p = 0; p < o->P; ++p) {
    /* AvatarConnector:postStart (avatar_library.lua:907-921): set the state
     * first (a piece without a layer cannot be teleported), teleport onto the
     * avatar, connect. */
    const Piece* av = &o->pieces[o->avatar_piece[p]];
    eng_set_state(o, c->mark_piece[p], c->s_ind[IND_NOT_READY]);
    eng_teleport(o, c->mark_piece[p], av->x, av->y);
    eng_set_orientation(o, c->mark_piece[p], av->orient);
    eng_connect(o, o->avatar_piece[p], c->mark_piece[p]);
  }
}

/* BaseSimulation:update (base_simulation.lua:476-486): preUpdate on all, then
 * update on all. */
static void mx_sim_update(Oracle* o) {
  Matrix* c = mx(o);
  for (int p = 0; p < o->P; ++p) o->reward[p] = 0.0; /* Avatar:preUpdate */
  c->ee_t++; /* StochasticIntervalEpisodeEnding:update */
  for (int p = 0; p < o->P; ++p) {
    /* Avatar:update (avatar_library.lua:334-355) */
    if (o->freeze_counter[p] == 1) o->movement_allowed[p] = 1;
    if (o->freeze_counter[p] > 0) o->freeze_counter[p]--;
    if (o->removal_counter[p] == 1) eng_set_state(o, o->avatar_piece[p], o->wait_state[p]);
    if (o->removal_counter[p] > 0) o->removal_counter[p]--;
    /* GameInteractionZapper:update (components.lua:899-906) */
    for (int s = 0; s < 2; ++s)
      for (int k = 0; k < c->R; ++k) c->latest[p][s][k] = -1.0;
    reset_cumulants(c, p);
  }
}

/* TheMatrix:getColorInterval (components.lua:282-290) */
static int color_interval(const Matrix* c, double reward) {
  for (int k = 0; k < c->n_intervals; ++k)
    if (c->interval[k][0] <= reward && reward < c->interval[k][1]) return k;
  abort(); /* the reference asserts */
}

/* InteractionTaste:getExtraRewardForInteraction (components.lua:1019-1039), the
 * component of player `owner`; `inventory` is read when the effect is applied. */
static double interaction_taste(const Matrix* c, int owner, double reward, const double* inventory) {
  const int tasty = c->itaste_class[owner];
  if (tasty > 0) {
    if (c->itaste_zero[owner]) reward = 0.0;
    double amount = inventory[tasty - 1];
    int maximal = 1;
    for (int idx = 1; idx <= c->R; ++idx)
      if (idx != tasty) maximal = amount > inventory[idx - 1]; /* (the last one decides, as written) */
    if (maximal) return reward + c->itaste_extra[owner];
  }
  return reward;
}

/* the scheduled effects of one interaction, in the order _resolve inserts them
 * (components.lua:634-693); `owner` = the zapped player, whose component holds
 * the list (and whose InteractionTaste prices both rewards, :527-549) */
static void apply_effects(Oracle* o, Matrix* c, int owner) {
  Effects* e = &c->fx[owner];
  if (!e->pending) return;
  e->pending = 0;
  const int row = e->row, col = e->col;
  /* sendRewardsToBothInteractants */
  if (e->row_reward > c->reward_floor)
    add_reward(o, row, interaction_taste(c, owner, e->row_reward, c->inv[row]));
  if (e->col_reward > c->reward_floor)
    add_reward(o, col, interaction_taste(c, owner, e->col_reward, c->inv[col]));
  const int winner = e->row_won ? row : col, loser = e->row_won ? col : row;
  if (c->reset_loser) reset_inventory(c, loser);
  if (c->reset_winner) reset_inventory(c, winner);
  /* _avatarDies */
  if (c->loser_dies) eng_set_state(o, o->avatar_piece[loser], o->wait_state[loser]);
  if (c->winner_dies) eng_set_state(o, o->avatar_piece[winner], o->wait_state[winner]);
}

static void mx_run_updaters(Oracle* o) {
  Matrix* c = mx(o);
  int order[ORC_MAX_PLAYERS];
  const int P = o->P;
  /* 900: GameInteractionZapper endEpisodeIfApplicable (components.lua:452-463) */
  eng_trace(o, 900, "GameInteractionZapper.endEpisodeIfApplicable");
  for (int p = 0; p < P; ++p)
    if (c->end_next_frame[p]) o->continue_flag = 0;
  /* 890: resetSimultaneousInteractionBlocker (:465-472) */
  eng_trace(o, 890, "GameInteractionZapper.resetSimultaneousInteractionBlocker");
  for (int p = 0; p < P; ++p) c->interacted_flag[p] = 0;
  /* 150: Avatar move (avatar_library.lua:155-203): turn self + connected, move */
  eng_trace(o, 150, "Avatar.move");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_MOVE, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i];
    if (!o->movement_allowed[p]) continue;
    int turn = o->action[p][ACT_TURNA], move = o->action[p][ACT_MOVE];
    if (turn != 0) {
      eng_turn(o, o->avatar_piece[p], turn);
      if (o->pieces[c->mark_piece[p]].leader == o->avatar_piece[p])
        eng_turn(o, c->mark_piece[p], turn);
    }
    if (move != 0) eng_move_rel(o, o->avatar_piece[p], move - 1);
  }
  /* 140: GameInteractionZapper zap (components.lua:400-424): nothing happens —
   * the cooling timer does not run either — while movement is disallowed */
  eng_trace(o, 140, "GameInteractionZapper.zap");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_ZAP, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i];
    if (!o->movement_allowed[p] || !is_alive(o, p) || c->cooldown < 0) continue;
    if (o->zap_timer[p] > 0) o->zap_timer[p]--;
    else if (o->action[p][ACT_INTERACT] == 1) { /* (_canZap: only DisallowMovement clears it) */
      o->zap_timer[p] = c->cooldown;
      eng_hit_beam(o, o->avatar_piece[p], c->hit, c->beam_length, c->beam_radius);
    }
  }
  /* 135: respawn, state = waitState, startFrame = framesTillRespawn (:426-437) */
  eng_trace(o, 135, "GameInteractionZapper.respawn");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_RESPAWN, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i], piece = o->avatar_piece[p];
    if (o->pieces[piece].state != o->wait_state[p]) continue;
    if (eng_frames(o, piece) < c->respawn_frames) continue;
    eng_teleport_to_group(o, piece, (uint32_t)o->spawn_group_mask, o->alive_state[p],
                          TELEPORT_PICK_RANDOM, RS_RESPAWN, p);
  }
  /* 100: StochasticIntervalEpisodeEnding (component_library.lua:927-940) */
  if (c->has_ee) {
    eng_trace(o, 100, "StochasticIntervalEpisodeEnding.maybeEndEpisode");
    if (eng_frames(o, 0) >= c->ee_min_frames && c->ee_t % c->ee_interval == 0)
      if (eng_u53(o, eng_draw(o, RS_EPISODE_END, 0)) < c->thr_ee) o->continue_flag = 0;
  }
  /* 100: Resource maybeRespawn (components.lua:84-101): state = waitState,
   * startFrame = regenerationDelay; the draw is the function's own
   * (uniformReal(0, 1) < regenerationRate), then only if no avatar stands on
   * the cell NOW (before this frame's moves) */
  eng_trace(o, 100, "Resource.maybeRespawn");
  for (int i = 0; i < c->n_site; ++i) {
    int piece = c->site_piece[i];
    if (piece < 0) continue;
    const Piece* r = &o->pieces[piece];
    if (r->state != c->s_wait[c->site_class[i]]) continue;
    if (eng_frames(o, piece) < c->regen_delay) continue;
    if (eng_u53(o, eng_draw(o, RS_REGROW, (uint32_t)i)) >= c->thr_regen) continue;
    if (eng_cell(o, o->avatar_layer, r->x, r->y) >= 0) continue;
    eng_set_state(o, piece, c->s_visible[c->site_class[i]]);
  }
  /* 7: SpawnResourcesWhenAllPlayersZapped (components.lua:303-321), one updater
   * per avatar object (idempotent within a frame: A2b) */
  if (c->spawn_all) {
    eng_trace(o, 7, "SpawnResourcesWhenAllPlayersZapped.step");
    int live = 0;
    for (int p = 0; p < P; ++p) live += is_alive(o, p);
    if (live == 0)
      for (int i = 0; i < c->n_site; ++i) {
        int piece = c->site_piece[i];
        if (piece >= 0 && o->pieces[piece].state == c->s_wait[c->site_class[i]])
          eng_set_state(o, piece, c->s_visible[c->site_class[i]]);
      }
  }
  /* 4: applyScheduledEffects, state = aliveState (components.lua:439-450) */
  eng_trace(o, 4, "GameInteractionZapper.applyScheduledEffects");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_CLEAN, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i];
    if (!is_alive(o, p)) continue;
    if (c->till_effects[p] == 0) {
      apply_effects(o, c, p);
      c->indicator[p] = IND_NOT_READY;
      c->till_effects[p] = -1;
      if (c->end_on_first) c->end_next_frame[p] = 1;
    } else if (c->till_effects[p] > 0) {
      c->till_effects[p]--;
      c->indicator[p] = IND_COLOR1 + color_interval(c, c->color_reward[p]);
    }
  }
  /* 2: ReadyToInteractMarker displayReadiness (components.lua:1081-1096) */
  eng_trace(o, 2, "ReadyToInteractMarker.displayReadiness");
  for (int p = 0; p < P; ++p) {
    if (is_alive(o, p)) eng_set_state(o, c->mark_piece[p], c->s_ind[c->indicator[p]]);
    else if (is_wait(o, p)) eng_set_state(o, c->mark_piece[p], c->s_mark_wait);
  }
}

/* GameInteractionZapper:_resolve (components.lua:556-703); `zapped` holds the
 * scheduled effects. */
static void resolve(Oracle* o, Matrix* c, int row, int col, int zapped, int hitter) {
  double row_profile[MX_MAX_R], col_profile[MX_MAX_R];
  double row_sum = 0.0, col_sum = 0.0;
  const int R = c->R;
  for (int k = 0; k < R; ++k) { row_sum += c->inv[row][k]; col_sum += c->inv[col][k]; }
  for (int k = 0; k < R; ++k) {
    row_profile[k] = row_sum > 0.0 ? c->inv[row][k] / row_sum : c->inv[row][k];
    col_profile[k] = col_sum > 0.0 ? c->inv[col][k] / col_sum : c->inv[col][k];
  }
  /* _computeInteractionRewards: (rowProfile * M) * colProfile, left to right */
  double row_reward = 0.0, col_reward = 0.0;
  {
    double tr[MX_MAX_R], tc[MX_MAX_R];
    for (int j = 0; j < R; ++j) {
      double a = 0.0, b = 0.0;
      for (int i = 0; i < R; ++i) {
        a += row_profile[i] * c->row_matrix[i][j];
        b += row_profile[i] * c->col_matrix[i][j];
      }
      tr[j] = a; tc[j] = b;
    }
    for (int j = 0; j < R; ++j) { row_reward += tr[j] * col_profile[j]; col_reward += tc[j] * col_profile[j]; }
  }
  row_reward = c->reward_multiplier * row_reward;
  col_reward = c->reward_multiplier * col_reward;
  /* reportInteraction on the zapped and on the zapper (:761-783): self first */
  const int both[2] = {zapped, hitter};
  for (int s = 0; s < 2; ++s) {
    int self = both[s], self_is_row = self == row;
    for (int k = 0; k < R; ++k) {
      c->latest[self][0][k] = self_is_row ? c->inv[row][k] : c->inv[col][k];
      c->latest[self][1][k] = self_is_row ? c->inv[col][k] : c->inv[row][k];
    }
  }
  /* reportEventAndCumulants (:785-806) */
  eng_event(o, 11 /* interaction (:790) */, row + 1, col + 1);
  c->latest_rewards[row][0] = c->latest_rewards[col][0] = row_reward;
  c->latest_rewards[row][1] = c->latest_rewards[col][1] = col_reward;
  const int rc[2] = {row, col};
  for (int s = 0; s < 2; ++s) { /* setArgMaxCumulants (:808-815): first maximal class */
    const double* inv = c->inv[rc[s]];
    int arg = 0;
    for (int k = 1; k < R; ++k) if (inv[k] > inv[arg]) arg = k;
    if (inv[arg] > 0.0) c->cum[rc[s]][3 + 3 * arg] = 1;
  }
  int row_won;
  if (row_reward > col_reward) row_won = 1;
  else if (row_reward == col_reward) {
    row_won = 1;
    if (c->random_tie) /* uniformReal(0, 1) <= 0.5 */
      row_won = eng_u53(o, eng_draw(o, RS_TIE_BREAK, (uint32_t)zapped)) <= ((uint64_t)1 << 52);
  } else row_won = 0;
  c->till_effects[row] = c->freeze; c->till_effects[col] = c->freeze;
  Effects* e = &c->fx[zapped];
  e->pending = 1; e->row = row; e->col = col; e->row_won = row_won;
  e->row_reward = row_reward; e->col_reward = col_reward;
  /* as written (:648-651): when the row player wins and reset_winner_inventory
   * is set, its inventory is ALSO reset right away, not only as an effect */
  if (row_won && c->reset_winner) reset_inventory(c, row);
  /* there is always at least the reward effect: freeze both (:695-702) */
  const int nfreeze = c->freeze + 2;
  if (nfreeze > 0) {
    o->movement_allowed[row] = 0; o->freeze_counter[row] = nfreeze;
    o->movement_allowed[col] = 0; o->freeze_counter[col] = nfreeze;
  }
  c->color_reward[row] = row_reward;
  c->color_reward[col] = col_reward;
}

static int mx_on_hit(Oracle* o, int target, int hitter, int hit) {
  Matrix* c = mx(o);
  const Piece* t = &o->pieces[target];
  if (hit != c->hit) return 0;
  if (c->state_hit_block[t->state] & (1u << hit)) return 1; /* BeamBlocker walls */
  const int hp = o->pieces[hitter].index; /* the zapper */
  if (t->kind == MPK_KIND_RESOURCE) {
    /* Destroyable:onHit (components.lua:154-172) */
    int i = t->index;
    c->health[i]--;
    if (c->health[i] == 0) {
      c->health[i] = c->initial_health;
      eng_set_state(o, target, c->s_wait[c->site_class[i]]);
      eng_event(o, 5 /* destroyed_resource (:178) */, hp + 1, c->site_class[i] + 1);
      c->cum[hp][2 + 3 * c->site_class[i]] = 1; /* setResourceDestructionCumulant */
      return 0; /* beams pass through a destroyed destroyable */
    }
    return 1;
  }
  if (t->kind != MPK_KIND_AVATAR) return 0;
  /* GameInteractionZapper:onHit (components.lua:720-759) */
  const int zp = t->index; /* the zapped */
  /* _preventExtraSimultaneousInteraction (:705-718) */
  if (c->interacted_flag[zp]) return 1;
  c->interacted_flag[zp] = 1;
  if (c->interacted_flag[hp]) return 1;
  c->interacted_flag[hp] = 1;
  if (c->till_effects[zp] >= 0) return 1; /* frozen players cannot be zapped */
  if (!c->collected[zp]) add_reward(o, hp, c->reward_unready);
  if (c->disallow_unready && !(c->collected[hp] && c->collected[zp])) return 1;
  c->cum[hp][0] = 1; c->cum[zp][0] = 1; /* _setInteractionCumulant */
  if (c->role[zp] >= 0 && c->role[hp] >= 0) { /* DyadicRole on both (:736-750) */
    if (c->role[hp] == 1 && c->role[zp] == 0) resolve(o, c, hp, zp, zp, hp);
    else if (c->role[hp] == 0 && c->role[zp] == 1) resolve(o, c, zp, hp, zp, hp);
  } else {
    resolve(o, c, hp, zp, zp, hp); /* the zapper is the row player */
  }
  return 1;
}

static void mx_on_enter(Oracle* o, int target, int entering, int contact) {
  Matrix* c = mx(o);
  (void)contact; /* the only contact is 'avatar' */
  const Piece* t = &o->pieces[target];
  if (t->kind != MPK_KIND_RESOURCE) return;
  /* Resource:onEnter (components.lua:54-82) */
  const int i = t->index, k = c->site_class[i];
  if (t->state != c->s_visible[k]) return;
  const int p = o->pieces[entering].index;
  c->inv[p][k] += 1.0;
  c->collected[p] = 1;
  if (c->indicator[p] == IND_NOT_READY) c->indicator[p] = IND_READY;
  eng_set_state(o, target, c->s_wait[k]);
  /* Taste:getRewardForGathering (:985-990) */
  add_reward(o, p, k + 1 == c->taste_class[p] ? c->taste_reward[p] : c->taste_default[p]);
  eng_event(o, 12 /* collected_resource (:117) */, p + 1, k + 1);
  c->cum[p][1 + 3 * k] = 1; /* setResourceCollectionCumulant */
}

static void mx_on_state_change(Oracle* o, int piece, int old_state) {
  Matrix* c = mx(o);
  const Piece* p = &o->pieces[piece];
  if (p->kind != MPK_KIND_AVATAR) return;
  const int pl = p->index; /* Avatar:onStateChange (avatar_library.lua:430-453) */
  const int mark = c->mark_piece[pl];
  if (old_state == o->wait_state[pl] && p->state == o->alive_state[pl]) {
    o->freeze_counter[pl] = 0; o->removal_counter[pl] = 0;
    /* 'respawn' -> AvatarConnector:avatarStateChange (avatar_library.lua:923-934) */
    eng_disconnect(o, mark);
    eng_set_state(o, mark, c->s_ind[IND_NOT_READY]);
    eng_teleport(o, mark, p->x, p->y);
    eng_set_orientation(o, mark, p->orient);
    eng_connect(o, piece, mark);
  } else if (old_state == o->alive_state[pl] && p->state == o->wait_state[pl]) {
    eng_set_state(o, mark, c->s_mark_wait); /* 'die' (:935-936) */
  }
}

const SubstrateVtbl kMatrixVtbl = {
    mx_on_enter, mx_on_hit, mx_on_state_change,
    mx_sim_update, mx_run_updaters, mx_start,
};
