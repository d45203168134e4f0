/* ORACLE — test infrastructure only.  See engine.h.
 *
 * commons_harvest rules: restatement of the reference's Lua components
 *   lua/levels/commons_harvest/components.lua  (Neighborhoods, DensityRegrow)
 *   lua/modules/component_library.lua:953-1004 (Edible),
 *                                    :907-948  (StochasticIntervalEpisodeEnding),
 *                                    :667-685  (BeamBlocker)
 *   lua/modules/avatar_library.lua             (Avatar, Zapper)
 * with kwargs from configs/substrates/commons_harvest__open.py (in the pack).
 */
#include <stdlib.h>
#include <string.h>

#include "../include/mp_pack.h"
#include "engine.h"

enum { HIT_ZAP = 0 };
enum { ACT_MOVE = 0, ACT_TURNA = 1, ACT_FIRE_ZAP = 2 };

typedef struct {
  int n_apple;
  int* apple_piece;      /* apple pieces in creation order */
  int* grass_piece;      /* DensityRegrow._underlyingGrass (queryPosition('background')) */
  int* num_neighbors;    /* Neighborhoods.pieceToNumNeighbors, by apple index */
  int started;           /* DensityRegrow._started */
  int s_apple, s_wait, s_grass, s_dess, s_wait_k[32], nk;
  int live_layer, wait_layer;
  const int32_t* disc;   /* queryDisc offsets (dx, dy), self excluded */
  int ndisc;
  const uint64_t* thr;   /* regrowth threshold per wait group, then episode end */
  const uint32_t* state_hit_block;
  int zap_cooldown, zap_length, zap_radius, respawn_frames, remove_hit;
  double zap_penalty, zap_reward, eat_reward;
  int ee_min_frames, ee_interval, ee_t;
} Commons;

static Commons* ch(const Oracle* o) { return (Commons*)o->sub_state; }

void* commons_create(Oracle* o) {
  Commons* c = (Commons*)calloc(1, sizeof(Commons));
  uint64_t n;
  const int32_t* st = (const int32_t*)mpk_find(o->pack, "ch_states", &n, 0);
  const int32_t* ci = (const int32_t*)mpk_find(o->pack, "ch_i32", &n, 0);
  const double* cf = (const double*)mpk_find(o->pack, "ch_f64", &n, 0);
  c->s_apple = st[0]; c->s_wait = st[1]; c->s_grass = st[2]; c->s_dess = st[3];
  c->nk = ci[0];
  for (int k = 0; k < c->nk; ++k) c->s_wait_k[k] = st[4 + k];
  c->ee_min_frames = ci[1]; c->ee_interval = ci[2];
  c->eat_reward = cf[0];
  c->live_layer = o->state_layer[c->s_apple];
  c->wait_layer = o->state_layer[c->s_wait];
  c->disc = (const int32_t*)mpk_find(o->pack, "disc_offsets", &n, 0);
  c->ndisc = (int)(n / 2);
  c->thr = (const uint64_t*)mpk_find(o->pack, "ch_thr", &n, 0);
  c->state_hit_block = (const uint32_t*)mpk_find(o->pack, "state_hit_block", &n, 0);
  const int32_t* zi = (const int32_t*)mpk_find(o->pack, "zapper_i32", &n, 0);
  const double* zf = (const double*)mpk_find(o->pack, "zapper_f64", &n, 0);
  c->zap_cooldown = zi[0]; c->zap_length = zi[1]; c->zap_radius = zi[2];
  c->respawn_frames = zi[3]; c->remove_hit = zi[4];
  c->zap_penalty = zf[0]; c->zap_reward = zf[1];
  mpk_find(o->pack, "apple_cells", &n, 0);
  c->n_apple = (int)n;
  c->apple_piece = (int*)calloc((size_t)c->n_apple, sizeof(int));
  c->grass_piece = (int*)calloc((size_t)c->n_apple, sizeof(int));
  c->num_neighbors = (int*)calloc((size_t)c->n_apple, sizeof(int));
  return c;
}

void commons_destroy(void* s) {
  Commons* c = (Commons*)s;
  if (!c) return;
  free(c->apple_piece); free(c->grass_piece); free(c->num_neighbors); free(c);
}

int commons_live_apples(const Oracle* o) {
  const Commons* c = ch(o);
  int n = 0;
  for (int i = 0; i < c->n_apple; ++i) n += o->pieces[c->apple_piece[i]].state == c->s_apple;
  return n;
}

static int is_alive(const Oracle* o, int p) {
  return o->pieces[o->avatar_piece[p]].state == o->alive_state[p];
}
static void add_reward(Oracle* o, int p, double amount) {
  /* Avatar:addReward with skipWaitStateRewards (avatar_library.lua:362-376) */
  if (o->pieces[o->avatar_piece[p]].state != o->wait_state[p]) o->reward[p] += amount;
}

/* transform:queryDisc(layer, radius): the pieces on `layer` within the L2 disc
 * around `piece` (self's own cell included when it is on that layer). */
static int query_disc(const Oracle* o, const Commons* c, int piece, int layer, int* out) {
  const Piece* p = &o->pieces[piece];
  int n = 0;
  int self = eng_cell(o, layer, p->x, p->y);
  if (self >= 0) out[n++] = self;
  for (int i = 0; i < c->ndisc; ++i) {
    int x = p->x + c->disc[2 * i], y = p->y + c->disc[2 * i + 1];
    if (o->topology == 1) { x = ((x % o->W) + o->W) % o->W; y = ((y % o->H) + o->H) % o->H; }
    else if (x < 0 || x >= o->W || y < 0 || y >= o->H) continue;
    int q = eng_cell(o, layer, x, y);
    if (q >= 0) out[n++] = q;
  }
  return n;
}

/* DensityRegrow:_beginLive (components.lua:205-219) */
static void begin_live(Oracle* o, int piece) {
  Commons* c = ch(o);
  int nb[64];
  int n = query_disc(o, c, piece, c->wait_layer, nb);
  for (int i = 0; i < n; ++i)
    if (nb[i] != piece && o->pieces[nb[i]].kind == MPK_KIND_DENSITY_REGROW)
      c->num_neighbors[o->pieces[nb[i]].index]++;
}

/* DensityRegrow:_endLive (components.lua:221-240) */
static void end_live(Oracle* o, int piece) {
  Commons* c = ch(o);
  int wait_nb[64], live_nb[64];
  int nw = query_disc(o, c, piece, c->wait_layer, wait_nb);
  int nl = query_disc(o, c, piece, c->live_layer, live_nb);
  for (int i = 0; i < nw; ++i) {
    if (o->pieces[wait_nb[i]].kind != MPK_KIND_DENSITY_REGROW) continue;
    int idx = o->pieces[wait_nb[i]].index;
    if (wait_nb[i] != piece) c->num_neighbors[idx]--;
    else c->num_neighbors[idx] = nl; /* self: #liveNeighbors */
    if (c->num_neighbors[idx] < 0) abort(); /* 'Less than zero neighbors' */
  }
}

static void ch_start(Oracle* o) {
  Commons* c = ch(o);
  int na = 0;
  for (int i = 0; i < o->npieces; ++i)
    if (o->pieces[i].kind == MPK_KIND_DENSITY_REGROW) c->apple_piece[na++] = i;
  c->started = 0;   /* DensityRegrow:reset */
  c->ee_t = 1;
  /* DensityRegrow:start: pieceToNumNeighbors[piece] = 0 */
  memset(c->num_neighbors, 0, (size_t)c->n_apple * sizeof(int));
  /* DensityRegrow:postStart: _beginLive, _started = true, _underlyingGrass */
  int bg = o->state_layer[c->s_grass];
  for (int i = 0; i < c->n_apple; ++i) {
    const Piece* p = &o->pieces[c->apple_piece[i]];
    begin_live(o, c->apple_piece[i]);
    c->grass_piece[i] = eng_cell(o, bg, p->x, p->y);
  }
  c->started = 1;
}

/* BaseSimulation:update (base_simulation.lua:476-486): preUpdate on all, then
 * update on all, objects in creation order: scene, avatars, map objects. */
static void ch_sim_update(Oracle* o) {
  Commons* c = ch(o);
  for (int p = 0; p < o->P; ++p) o->reward[p] = 0.0; /* Avatar:preUpdate */
  c->ee_t++; /* StochasticIntervalEpisodeEnding:update */
  for (int p = 0; p < o->P; ++p) { /* Avatar:update (avatar_library.lua:334-355) */
    if (o->freeze_counter[p] == 1) o->movement_allowed[p] = 1;
    if (o->freeze_counter[p] > 0) o->freeze_counter[p]--;
    if (o->removal_counter[p] == 1) eng_set_state(o, o->avatar_piece[p], o->wait_state[p]);
    if (o->removal_counter[p] > 0) o->removal_counter[p]--;
  }
  /* DensityRegrow:update -> _updateWaitState (components.lua:161-193) */
  for (int i = 0; i < c->n_apple; ++i) {
    int piece = c->apple_piece[i];
    int state = o->pieces[piece].state;
    if (o->state_layer[state] != c->wait_layer) continue; /* getLayer() == 'logic' */
    if (state == c->s_apple) continue;
    int num_close = c->num_neighbors[i];
    if (num_close >= c->nk) abort(); /* no such state in the reference either */
    eng_set_state(o, piece, c->s_wait_k[num_close]);
    if (c->grass_piece[i] >= 0)
      eng_set_state(o, c->grass_piece[i], num_close == 0 ? c->s_dess : c->s_grass);
  }
}

static void ch_run_updaters(Oracle* o) {
  Commons* c = ch(o);
  int order[ORC_MAX_PLAYERS];
  const int P = o->P;
  /* 150: Avatar move (avatar_library.lua:155-203) */
  eng_trace(o, 150, "Avatar.move");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_MOVE, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i];
    if (!o->movement_allowed[p]) continue;
    int turn = o->action[p][ACT_TURNA], move = o->action[p][ACT_MOVE];
    if (turn != 0) eng_turn(o, o->avatar_piece[p], turn);
    if (move != 0) eng_move_rel(o, o->avatar_piece[p], move - 1);
  }
  /* 140: Zapper zap (avatar_library.lua:613-636) */
  eng_trace(o, 140, "Zapper.zap");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_ZAP, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i];
    if (!is_alive(o, p) || c->zap_cooldown < 0) continue;
    if (o->zap_timer[p] > 0) o->zap_timer[p]--;
    else if (o->action[p][ACT_FIRE_ZAP] == 1) {
      o->zap_timer[p] = c->zap_cooldown;
      eng_hit_beam(o, o->avatar_piece[p], HIT_ZAP, c->zap_length, c->zap_radius);
    }
  }
  /* 135: Zapper respawn (avatar_library.lua:638-649) */
  eng_trace(o, 135, "Zapper.respawn");
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
  eng_trace(o, 100, "StochasticIntervalEpisodeEnding.maybeEndEpisode");
  if (eng_frames(o, 0) >= c->ee_min_frames && c->ee_t % c->ee_interval == 0) {
    if (eng_u53(o, eng_draw(o, RS_EPISODE_END, 0)) < c->thr[c->nk]) o->continue_flag = 0;
  }
  /* 10: DensityRegrow sprout, one engine-side probabilistic updater per wait
   * group (components.lua:104-137).  A12: every piece of the group is selected
   * independently with the group's probability, one draw per piece. */
  eng_trace(o, 10, "DensityRegrow.sprout");
  for (int k = 0; k < c->nk; ++k)
    for (int i = 0; i < c->n_apple; ++i) {
      int piece = c->apple_piece[i];
      if (o->pieces[piece].state != c->s_wait_k[k]) continue;
      if (eng_u53(o, eng_draw(o, RS_REGROW, (uint32_t)i)) < c->thr[k])
        eng_set_state(o, piece, c->s_apple); /* canRegrowIfOccupied = true */
    }
}

static int ch_on_hit(Oracle* o, int target, int hitter, int hit) {
  Commons* c = ch(o);
  const Piece* t = &o->pieces[target];
  int blocked = 0;
  if (c->state_hit_block[t->state] & (1u << hit)) blocked = 1; /* BeamBlocker */
  if (t->kind == MPK_KIND_AVATAR && hit == HIT_ZAP) { /* Zapper:onHit */
    int zapped = o->pieces[target].index, zapper = o->pieces[hitter].index;
    eng_event(o, 1 /* zap (avatar_library.lua:661) */, zapper + 1, zapped + 1);
    o->zap_matrix[zapped][zapper]++; o->num_zapped[zapper]++;
    add_reward(o, zapped, c->zap_penalty);
    add_reward(o, zapper, c->zap_reward);
    if (c->remove_hit) eng_set_state(o, target, o->wait_state[zapped]);
    blocked = 1;
  }
  return blocked;
}

static void ch_on_enter(Oracle* o, int target, int entering, int contact) {
  Commons* c = ch(o);
  (void)contact; /* the only contact is 'avatar' */
  const Piece* t = &o->pieces[target];
  /* Edible:onEnter (component_library.lua:990-1004) */
  if (t->kind == MPK_KIND_DENSITY_REGROW && t->state == c->s_apple) {
    add_reward(o, o->pieces[entering].index, c->eat_reward);
    eng_event(o, 2 /* edible_consumed (component_library.lua:996) */,
              o->pieces[entering].index + 1, 0);
    eng_set_state(o, target, c->s_wait);
  }
}

static void ch_on_state_change(Oracle* o, int piece, int old_state) {
  Commons* c = ch(o);
  const Piece* p = &o->pieces[piece];
  if (p->kind == MPK_KIND_DENSITY_REGROW) {
    /* DensityRegrow:onStateChange (components.lua:149-159) */
    if (!c->started) return;
    if (p->state == c->s_apple) begin_live(o, piece);
    else if (old_state == c->s_apple) end_live(o, piece);
  } else if (p->kind == MPK_KIND_AVATAR) {
    int pl = p->index; /* Avatar:onStateChange (avatar_library.lua:430-453) */
    if (old_state == o->wait_state[pl] && p->state == o->alive_state[pl]) {
      o->freeze_counter[pl] = 0;
      o->removal_counter[pl] = 0;
    }
  }
}

const SubstrateVtbl kCommonsVtbl = {
    ch_on_enter, ch_on_hit, ch_on_state_change,
    ch_sim_update, ch_run_updaters, ch_start,
};
