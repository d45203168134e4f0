/* ORACLE — test infrastructure only.  See engine.h.
 *
 * externality_mushrooms rules: restatement of the reference's Lua components
 *   lua/levels/externality_mushrooms/components.lua  (MushroomEating :30-153,
 *       MushroomGrowable :155-194, MushroomRegrowth :196-255, Destroyable :257-306,
 *       Perishable :308-335; Cumulants :337-372 feeds debug observations only)
 *   lua/modules/avatar_library.lua   (Avatar, Zapper incl. respawn, the timed freeze / zap
 *       prevention / scheduled removal, GraduatedSanctionsMarking :948-1121 incl. its
 *       'respawn' branch :1099-1110)
 *   lua/modules/component_library.lua:907-948  (StochasticIntervalEpisodeEnding),
 *                                    :667-685  (BeamBlocker)
 * with kwargs from configs/substrates/externality_mushrooms.py + ..._dense.py (in the pack).
 *
 * A20: grid:createPiece fires the state's onAdd like a setState does, so every
 * component's onStateChange(nil) runs when BaseSimulation:start creates the pieces
 * (game_object.lua:262-273, :344).  It matters in this level only: MushroomGrowable marks
 * every mushroom created in 'wait' for registration as a potential site (and every mushroom
 * created live for a de-registration that only lowers the counter) — without it the set of
 * potential sites starts empty and the map of the "dense" variant, whose docstring says
 * mushrooms may grow anywhere, could only ever regrow its twelve initial sites.
 *
 * The Lua iterates `mushroomsToProbabilities[eaten]` with pairs() (components.lua:218): the
 * order is the pack's (config order; A11: any fixed order conforms).
 */
#include <stdlib.h>
#include <string.h>

#include "../include/mp_pack.h"
#include "engine.h"

enum { ACT_MOVE = 0, ACT_TURNA = 1, ACT_FIRE_ZAP = 2 };
enum { NT = 4 };   /* mushroom types: the four live states of the prefab */

typedef struct {
  int n_site;
  int* piece;                      /* mushroom pieces in creation order */
  uint8_t *in_set, *must_reg, *must_dereg;   /* MushroomRegrowth._potentialMushrooms; MushroomGrowable flags */
  int* health;                     /* Destroyable._variables.health */
  int num_potential;               /* MushroomRegrowth._numPotentialMushrooms */
  int s_type[NT], s_wait;
  int spores[NT], digest[NT], perish[NT], destroy_type[NT];
  double total_reward[NT];
  uint64_t grow_thr[NT][NT], destroy_thr[NT];
  int min_potential, initial_health;
  /* GraduatedSanctionsMarking / Zapper timed prevention (as territory.c) */
  int mark_piece[ORC_MAX_PLAYERS];
  int level[ORC_MAX_PLAYERS], time_since_not_initial[ORC_MAX_PLAYERS];
  int disallow_zapping[ORC_MAX_PLAYERS], no_zapping_counter[ORC_MAX_PLAYERS];
  int s_mark[2], s_mark_wait, recovery_time, nlevels;
  int lv_increment[2], lv_freeze[2], lv_remove[2];
  double lv_source[2], lv_target[2];
  int hit_zap, zap_cooldown, zap_length, zap_radius, respawn_frames, remove_hit;
  double zap_penalty, zap_reward;
  int ee_min_frames, ee_interval, ee_t;
  uint64_t thr_ee;
  const uint32_t* state_hit_block;
} Mush;

static Mush* mu(const Oracle* o) { return (Mush*)o->sub_state; }

void* mushroom_create(Oracle* o) {
  Mush* c = (Mush*)calloc(1, sizeof(Mush));
  uint64_t n;
  const int32_t* st = (const int32_t*)mpk_find(o->pack, "em_states", &n, 0);
  const int32_t* ci = (const int32_t*)mpk_find(o->pack, "em_i32", &n, 0);
  const double* cf = (const double*)mpk_find(o->pack, "em_f64", &n, 0);
  const uint64_t* thr = (const uint64_t*)mpk_find(o->pack, "em_thr", &n, 0);
  for (int k = 0; k < NT; ++k) c->s_type[k] = st[k];
  c->s_wait = st[4]; c->s_mark[0] = st[5]; c->s_mark[1] = st[6]; c->s_mark_wait = st[7];
  c->min_potential = ci[0]; c->initial_health = ci[1]; c->recovery_time = ci[2];
  c->nlevels = ci[3]; c->ee_min_frames = ci[4]; c->ee_interval = ci[5]; c->hit_zap = ci[6];
  if (c->nlevels > 2) abort();
  for (int k = 0; k < NT; ++k) {
    c->spores[k] = ci[8 + k]; c->digest[k] = ci[12 + k]; c->perish[k] = ci[16 + k];
    c->destroy_type[k] = ci[20 + k];
    c->total_reward[k] = cf[k];
    for (int m = 0; m < NT; ++m) c->grow_thr[k][m] = thr[k * NT + m];
    c->destroy_thr[k] = thr[NT * NT + k];
  }
  c->thr_ee = thr[NT * NT + NT];
  for (int l = 0; l < c->nlevels; ++l) {
    c->lv_increment[l] = ci[24 + 3 * l]; c->lv_freeze[l] = ci[25 + 3 * l];
    c->lv_remove[l] = ci[26 + 3 * l];
    c->lv_source[l] = cf[4 + 2 * l]; c->lv_target[l] = cf[5 + 2 * l];
  }
  const int32_t* zi = (const int32_t*)mpk_find(o->pack, "zapper_i32", &n, 0);
  const double* zf = (const double*)mpk_find(o->pack, "zapper_f64", &n, 0);
  c->zap_cooldown = zi[0]; c->zap_length = zi[1]; c->zap_radius = zi[2];
  c->respawn_frames = zi[3]; c->remove_hit = zi[4];
  c->zap_penalty = zf[0]; c->zap_reward = zf[1];
  c->state_hit_block = (const uint32_t*)mpk_find(o->pack, "state_hit_block", &n, 0);
  mpk_find(o->pack, "mushroom_cells", &n, 0);
  c->n_site = (int)n;
  c->piece = (int*)calloc((size_t)c->n_site, sizeof(int));
  c->health = (int*)calloc((size_t)c->n_site, sizeof(int));
  c->in_set = (uint8_t*)calloc((size_t)c->n_site, 1);
  c->must_reg = (uint8_t*)calloc((size_t)c->n_site, 1);
  c->must_dereg = (uint8_t*)calloc((size_t)c->n_site, 1);
  return c;
}

void mushroom_destroy(void* s) {
  Mush* c = (Mush*)s;
  if (!c) return;
  free(c->piece); free(c->health); free(c->in_set); free(c->must_reg); free(c->must_dereg);
  free(c);
}

/* Extra parity fields of the canonical dump (mirrored by mp_dump):
 *   avat[p][7] as territory_dump; glob[3] = live mushrooms, glob[5] = sum over live
 *   mushrooms of (type + 1) * min(frames in state, 255), glob[6] = potential sites + 1000
 *   (the Lua's counter, which runs below the size of its set by the number of mushrooms
 *   the map starts with), glob[7] = size of the set */
void mushroom_dump(const Oracle* o, int32_t* avat, int32_t* glob) {
  const Mush* c = mu(o);
  for (int p = 0; p < o->P; ++p)
    avat[8 * p + 7] = c->level[p] | (o->freeze_counter[p] << 4) |
                      (o->removal_counter[p] << 12) | (c->no_zapping_counter[p] << 16) |
                      (o->movement_allowed[p] << 24) | (c->disallow_zapping[p] << 25);
  int live = 0, ages = 0, in_set = 0;
  for (int i = 0; i < c->n_site; ++i) {
    const int s = o->pieces[c->piece[i]].state;
    for (int k = 0; k < NT; ++k)
      if (s == c->s_type[k]) {
        int f = eng_frames(o, c->piece[i]);
        live++; ages += (k + 1) * (f < 255 ? f : 255);
      }
    in_set += c->in_set[i];
  }
  glob[3] = live; glob[5] = ages; glob[6] = c->num_potential + 1000; glob[7] = in_set;
}

static int is_alive(const Oracle* o, int p) {
  return o->pieces[o->avatar_piece[p]].state == o->alive_state[p];
}
static int is_wait(const Oracle* o, int p) {
  return o->pieces[o->avatar_piece[p]].state == o->wait_state[p];
}
static void add_reward(Oracle* o, int p, double amount) {
  /* Avatar:addReward, skipWaitStateRewards = true (avatar_library.lua:362-376) */
  if (!is_wait(o, p)) o->reward[p] += amount;
}
static int type_of(const Mush* c, int state) {
  for (int k = 0; k < NT; ++k) if (state == c->s_type[k]) return k;
  return -1;
}
/* GraduatedSanctionsMarking:_setLevel (avatar_library.lua:1112-1121) */
static void set_level(Oracle* o, Mush* c, int p, int level) {
  eng_set_state(o, c->mark_piece[p], c->s_mark[level - 1]);
  eng_event(o, 8 /* set_sanctioning_level */, p + 1, level);
}

static void em_start(Oracle* o) {
  Mush* c = mu(o);
  int n = 0;
  for (int i = 0; i < o->npieces; ++i) {
    if (o->pieces[i].kind == MPK_KIND_MUSHROOM) c->piece[n++] = i;
    if (o->pieces[i].kind == MPK_KIND_MARKING) c->mark_piece[o->pieces[i].index] = i;
  }
  if (n != c->n_site) abort();
  /* MushroomRegrowth:reset, Destroyable:reset; then the pieces are created and (A20) their
   * onAdd fires: MushroomGrowable:onStateChange(nil) (components.lua:184-194) */
  c->num_potential = 0;
  for (int i = 0; i < n; ++i) {
    c->in_set[i] = 0; c->health[i] = c->initial_health;
    const int waiting = o->pieces[c->piece[i]].state == c->s_wait;
    c->must_reg[i] = (uint8_t)waiting; c->must_dereg[i] = (uint8_t)!waiting;
  }
  c->ee_t = 1;
  for (int p = 0; p < o->P; ++p) {
    c->level[p] = 1; c->time_since_not_initial[p] = 0;        /* GSM:reset */
    c->disallow_zapping[p] = 0; c->no_zapping_counter[p] = 0; /* Zapper:reset */
    /* GraduatedSanctionsMarking:postStart (avatar_library.lua:1034-1049) */
    const Piece* av = &o->pieces[o->avatar_piece[p]];
    set_level(o, c, p, c->level[p]);
    eng_teleport(o, c->mark_piece[p], av->x, av->y);
    eng_set_orientation(o, c->mark_piece[p], av->orient);
    eng_connect(o, o->avatar_piece[p], c->mark_piece[p]);
  }
}

/* BaseSimulation:update: preUpdate on all, then update on all, objects in creation order:
 * scene, (avatar, marking) pairs, map objects. */
static void em_sim_update(Oracle* o) {
  Mush* c = mu(o);
  for (int p = 0; p < o->P; ++p) o->reward[p] = 0.0; /* Avatar:preUpdate */
  c->ee_t++;
  for (int p = 0; p < o->P; ++p) {
    /* Avatar:update (avatar_library.lua:334-355) */
    if (o->freeze_counter[p] == 1) o->movement_allowed[p] = 1;
    if (o->freeze_counter[p] > 0) o->freeze_counter[p]--;
    if (o->removal_counter[p] == 1) eng_set_state(o, o->avatar_piece[p], o->wait_state[p]);
    if (o->removal_counter[p] > 0) o->removal_counter[p]--;
    /* Zapper:update (avatar_library.lua:713-726) */
    if (c->disallow_zapping[p]) o->zap_timer[p] = c->zap_cooldown + 1;
    int old = c->no_zapping_counter[p];
    if (c->no_zapping_counter[p] > 0) c->no_zapping_counter[p]--;
    if (old == 1) c->disallow_zapping[p] = 0;
  }
}

static void em_run_updaters(Oracle* o) {
  Mush* c = mu(o);
  int order[ORC_MAX_PLAYERS];
  const int P = o->P;
  /* 900: Cumulants resetCumulants (components.lua:361-370): debug observations only */
  eng_trace(o, 900, "Cumulants.resetCumulants");
  /* 500: MushroomGrowable registration (components.lua:164-182) */
  eng_trace(o, 500, "MushroomGrowable.registration");
  for (int i = 0; i < c->n_site; ++i) {
    if (c->must_reg[i]) { c->in_set[i] = 1; c->num_potential++; }          /* :246-249 */
    else if (c->must_dereg[i]) { c->in_set[i] = 0; c->num_potential--; }   /* :251-254 */
    c->must_reg[i] = c->must_dereg[i] = 0;
  }
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
      eng_turn(o, c->mark_piece[p], turn);
    }
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
      eng_hit_beam(o, o->avatar_piece[p], c->hit_zap, c->zap_length, c->zap_radius);
    }
  }
  /* 135: Zapper respawn, state = waitState, startFrame = framesTillRespawn (:638-649) */
  eng_trace(o, 135, "Zapper.respawn");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_RESPAWN, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i], piece = o->avatar_piece[p];
    if (is_wait(o, p) && eng_frames(o, piece) >= c->respawn_frames)
      eng_teleport_to_group(o, piece, (uint32_t)o->spawn_group_mask, o->alive_state[p],
                            TELEPORT_PICK_RANDOM, RS_RESPAWN, p);
  }
  /* 100: StochasticIntervalEpisodeEnding */
  eng_trace(o, 100, "StochasticIntervalEpisodeEnding.maybeEndEpisode");
  if (eng_frames(o, 0) >= c->ee_min_frames && c->ee_t % c->ee_interval == 0)
    if (eng_u53(o, eng_draw(o, RS_EPISODE_END, 0)) < c->thr_ee) o->continue_flag = 0;
  /* 3: GraduatedSanctionsMarking resetToInitialLevel (avatar_library.lua:1010-1026) */
  eng_trace(o, 3, "GraduatedSanctionsMarking.resetToInitialLevel");
  for (int p = 0; p < P; ++p) {
    if (c->level[p] != 1 && is_alive(o, p)) {
      c->time_since_not_initial[p]++;
      if (c->time_since_not_initial[p] == c->recovery_time) {
        c->level[p] = 1;
        set_level(o, c, p, 1);
        c->time_since_not_initial[p] = 0;
      }
    }
  }
  /* 3: Perishable perish, one updater per live state, startFrame = its delay (:321-335) */
  for (int k = 0; k < NT; ++k) {
    static const char* tags[NT] = {"Perishable.perish_0", "Perishable.perish_1",
                                   "Perishable.perish_2", "Perishable.perish_3"};
    eng_trace(o, 3, tags[k]);
    for (int i = 0; i < c->n_site; ++i) {
      const int piece = c->piece[i];
      if (o->pieces[piece].state == c->s_type[k] && eng_frames(o, piece) >= c->perish[k])
        eng_set_state(o, piece, c->s_wait);
    }
  }
}

/* MushroomRegrowth:grow (components.lua:216-235); `draw_index` names this call's draws */
static void grow(Oracle* o, Mush* c, int eaten, uint32_t draw_index) {
  for (int m = 0; m < NT; ++m) {
    if (c->num_potential < c->min_potential) continue;
    PhiloxOut d = eng_draw(o, RS_MUSHROOM_GROW, draw_index * NT + (uint32_t)m);
    if (eng_u53(o, d) >= c->grow_thr[eaten][m]) continue;
    /* random:choice(set.toSortedList(potentials)): pieces sort in creation order */
    int n = 0;
    for (int i = 0; i < c->n_site; ++i) n += c->in_set[i];
    if (n == 0) continue;
    int k = (int)eng_bounded(o, d, (uint32_t)n), site = -1;
    for (int i = 0; i < c->n_site; ++i)
      if (c->in_set[i] && k-- == 0) { site = i; break; }
    const Piece* pc = &o->pieces[c->piece[site]];
    if (eng_cell(o, o->avatar_layer, pc->x, pc->y) >= 0) continue;   /* queryPosition('upperPhysical') */
    eng_set_state(o, c->piece[site], c->s_type[m]);
  }
}

/* MushroomEating:onEnter (components.lua:107-138) */
static void em_on_enter(Oracle* o, int target, int entering, int contact) {
  Mush* c = mu(o);
  (void)contact; /* the only contact in this level is 'avatar' */
  const Piece* t = &o->pieces[target];
  const Piece* e = &o->pieces[entering];
  if (t->kind != MPK_KIND_MUSHROOM || e->kind != MPK_KIND_AVATAR) return;
  const int type = type_of(c, t->state);
  if (type < 0) return;
  const int p = e->index, P = o->P;
  /* _rewardEveryone (:65-105); the type's name picks the rule, the pack its number */
  eng_event(o, 20 /* eating_mushroom */, p + 1, type + 1);
  if (type == 0) {
    add_reward(o, p, c->total_reward[0]);
  } else if (type == 1 || type == 3) {
    const double part = c->total_reward[type] / (double)P;
    add_reward(o, p, part);
    for (int q = 0; q < P; ++q) if (q != p) add_reward(o, q, part);
  } else {
    const double part = c->total_reward[2] / (double)(P - 1);
    for (int q = 0; q < P; ++q) if (q != p) add_reward(o, q, part);
  }
  for (int n = 0; n < c->spores[type]; ++n) grow(o, c, type, (uint32_t)(p * 4 + n));
  if (c->destroy_type[type] >= 0) {
    /* getGroupShuffledWithProbability(typeToDestroy, percentToDestroy): every piece of the
     * group with that probability; all of them get the same setState */
    const int victim = c->s_type[c->destroy_type[type]];
    for (int i = 0; i < c->n_site; ++i)
      if (o->pieces[c->piece[i]].state == victim &&
          eng_u53(o, eng_draw(o, RS_MUSHROOM_DESTROY, (uint32_t)(p * 256 + i))) < c->destroy_thr[type])
        eng_set_state(o, c->piece[i], c->s_wait);
  }
  if (c->digest[type] > 0) {   /* Avatar:disallowMovementUntil */
    o->movement_allowed[p] = 0; o->freeze_counter[p] = c->digest[type];
  }
  eng_set_state(o, target, c->s_wait);
}

static int em_on_hit(Oracle* o, int target, int hitter, int hit) {
  Mush* c = mu(o);
  const Piece* t = &o->pieces[target];
  int blocked = 0;
  if (c->state_hit_block[t->state] & (1u << hit)) blocked = 1; /* BeamBlocker */
  if (hit != c->hit_zap) return blocked;
  const int hp = o->pieces[hitter].index;
  if (t->kind == MPK_KIND_AVATAR) {
    /* Zapper:onHit (avatar_library.lua:652-681) */
    eng_event(o, 1 /* zap */, hp + 1, t->index + 1);
    add_reward(o, t->index, c->zap_penalty);
    add_reward(o, hp, c->zap_reward);
    if (c->remove_hit) eng_set_state(o, target, o->wait_state[t->index]);
    blocked = 1;
  } else if (t->kind == MPK_KIND_MARKING) {
    /* GraduatedSanctionsMarking:onHit (avatar_library.lua:1051-1097) */
    int p = t->index, l = c->level[p] - 1;
    add_reward(o, hp, c->lv_source[l]);
    add_reward(o, p, c->lv_target[l]);
    c->level[p] += c->lv_increment[l];
    if (c->lv_remove[l]) {
      o->removal_counter[p] = 1;                            /* removeAfterDelay(1) */
      o->movement_allowed[p] = 0; o->freeze_counter[p] = 1; /* disallowMovementUntil(1) */
      c->disallow_zapping[p] = 1; c->no_zapping_counter[p] = 1;
      eng_event(o, 7 /* removal_due_to_sanctioning */, hp + 1, p + 1);
    } else {
      set_level(o, c, p, c->level[p]);
      if (c->lv_freeze[l] > 0) {
        o->movement_allowed[p] = 0; o->freeze_counter[p] = c->lv_freeze[l];
        c->disallow_zapping[p] = 1; c->no_zapping_counter[p] = c->lv_freeze[l];
      }
    }
    c->time_since_not_initial[p] = 0;
    eng_event(o, 6 /* sanctioning */, hp + 1, p + 1);
  } else if (t->kind == MPK_KIND_MUSHROOM) {
    /* Destroyable:onHit (components.lua:275-291) */
    const int i = t->index;
    c->health[i]--;
    if (c->health[i] == 0) {
      c->health[i] = c->initial_health;
      eng_set_state(o, target, c->s_wait);
      return blocked;   /* beams pass a destroyed destroyable */
    }
    blocked = 1;
  }
  return blocked;
}

static void em_on_state_change(Oracle* o, int piece, int old_state) {
  Mush* c = mu(o);
  const Piece* p = &o->pieces[piece];
  if (p->kind == MPK_KIND_MUSHROOM) {
    /* MushroomGrowable:onStateChange (components.lua:184-194) */
    if (p->state == c->s_wait) c->must_reg[p->index] = 1;
    else c->must_dereg[p->index] = 1;
    return;
  }
  if (p->kind != MPK_KIND_AVATAR) return;
  int pl = p->index; /* Avatar:onStateChange (avatar_library.lua:430-453) */
  if (old_state == o->wait_state[pl] && p->state == o->alive_state[pl]) {
    o->freeze_counter[pl] = 0; o->removal_counter[pl] = 0;
    /* 'respawn' -> GraduatedSanctionsMarking:avatarStateChange (:1099-1110) */
    eng_disconnect(o, c->mark_piece[pl]);
    set_level(o, c, pl, c->level[pl]);
    eng_teleport(o, c->mark_piece[pl], p->x, p->y);
    eng_set_orientation(o, c->mark_piece[pl], p->orient);
    eng_connect(o, piece, c->mark_piece[pl]);
  } else if (old_state == o->alive_state[pl] && p->state == o->wait_state[pl]) {
    eng_set_state(o, c->mark_piece[pl], c->s_mark_wait);   /* 'die' */
  }
}

const SubstrateVtbl kMushroomVtbl = {
    em_on_enter, em_on_hit, em_on_state_change,
    em_sim_update, em_run_updaters, em_start,
};
