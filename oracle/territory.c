/* ORACLE — test infrastructure only.  See engine.h.
 *
 * territory rules: restatement of the reference's Lua components
 *   lua/levels/territory/components.lua  (AllBeamBlocker, Resource,
 *     ResourceClaimer, RewardIndicator, Taste (role 'none'), Paintbrush)
 *   lua/levels/territory/init.lua        (two extra render layers)
 *   lua/modules/avatar_library.lua       (Avatar, Zapper incl. the timed
 *     freeze / zap prevention / scheduled removal, GraduatedSanctionsMarking)
 *   lua/modules/component_library.lua    (StochasticIntervalEpisodeEnding)
 * with kwargs from configs/substrates/territory.py + territory__rooms.py.
 */
#include <stdlib.h>
#include <string.h>

#include "../include/mp_pack.h"
#include "engine.h"

enum { ACT_MOVE = 0, ACT_TURNA = 1, ACT_FIRE_ZAP = 2, ACT_FIRE_CLAIM = 3 };

typedef struct {
  int P, n_res;
  int *res_piece, *tex_piece, *ind_piece, *dmg_piece; /* per resource, creation order */
  int mark_piece[ORC_MAX_PLAYERS];
  /* Resource volatile variables (territory/components.lua:75-82) */
  int *health, *frames_since_zapped, *active, *claimed_by, *destroyed;
  /* GraduatedSanctionsMarking (avatar_library.lua:994-1008) */
  int level[ORC_MAX_PLAYERS], time_since_not_initial[ORC_MAX_PLAYERS];
  /* Zapper timed prevention (avatar_library.lua:687-691) */
  int disallow_zapping[ORC_MAX_PLAYERS], no_zapping_counter[ORC_MAX_PLAYERS];
  /* ResourceClaimer._cooldown */
  int claim_cooldown[ORC_MAX_PLAYERS];
  int ee_t;
  /* states */
  int s_res_unclaimed, s_res_destroyed, s_tex_destroyed, s_ind_inactive, s_dmg_inactive,
      s_dmg_damaged, s_mark[2], s_mark_wait, s_claimed[ORC_MAX_PLAYERS],
      s_dry[ORC_MAX_PLAYERS];
  /* constants */
  int initial_health, reward_delay, repair_delay, claim_length, claim_radius, claim_wait,
      recovery_time, nlevels, ee_min_frames, ee_interval;
  int lv_increment[4], lv_freeze[4], lv_remove[4];
  double reward, lv_source[4], lv_target[4];
  uint64_t thr_reward, thr_repair, thr_ee;
  int hit_zap, hit_brush[ORC_MAX_PLAYERS], hit_claim[ORC_MAX_PLAYERS];
  int zap_cooldown, zap_length, zap_radius, respawn_frames, remove_hit;
  double zap_penalty, zap_reward;
  const uint32_t* state_hit_block;
} Territory;

static Territory* tr(const Oracle* o) { return (Territory*)o->sub_state; }

void* territory_create(Oracle* o) {
  Territory* c = (Territory*)calloc(1, sizeof(Territory));
  uint64_t n;
  c->P = o->P;
  const int PP = o->P_pack; /* table strides */
  const int32_t* st = (const int32_t*)mpk_find(o->pack, "tr_states", &n, 0);
  const int32_t* ci = (const int32_t*)mpk_find(o->pack, "tr_i32", &n, 0);
  const double* cf = (const double*)mpk_find(o->pack, "tr_f64", &n, 0);
  const uint64_t* thr = (const uint64_t*)mpk_find(o->pack, "tr_thr", &n, 0);
  const int32_t* hits = (const int32_t*)mpk_find(o->pack, "tr_hits", &n, 0);
  c->s_res_unclaimed = st[0]; c->s_res_destroyed = st[1]; c->s_tex_destroyed = st[3];
  c->s_ind_inactive = st[4]; c->s_dmg_inactive = st[5]; c->s_dmg_damaged = st[6];
  c->s_mark[0] = st[7]; c->s_mark[1] = st[8]; c->s_mark_wait = st[9];
  for (int p = 0; p < PP; ++p) { c->s_claimed[p] = st[10 + p]; c->s_dry[p] = st[10 + PP + p]; }
  c->initial_health = ci[0]; c->reward_delay = ci[1]; c->repair_delay = ci[2];
  c->claim_length = ci[3]; c->claim_radius = ci[4]; c->claim_wait = ci[5];
  c->recovery_time = ci[6]; c->nlevels = ci[7]; c->ee_min_frames = ci[8]; c->ee_interval = ci[9];
  if (c->nlevels > 2) abort(); /* two marking states are lowered */
  for (int l = 0; l < c->nlevels; ++l) {
    c->lv_increment[l] = ci[10 + 3 * l]; c->lv_freeze[l] = ci[11 + 3 * l];
    c->lv_remove[l] = ci[12 + 3 * l];
    c->lv_source[l] = cf[4 + 2 * l]; c->lv_target[l] = cf[5 + 2 * l];
  }
  c->reward = cf[0];
  c->thr_reward = thr[0]; c->thr_repair = thr[1]; c->thr_ee = thr[2];
  c->hit_zap = hits[0];
  for (int p = 0; p < PP; ++p) { c->hit_brush[p] = hits[1 + p]; c->hit_claim[p] = hits[1 + PP + p]; }
  const int32_t* zi = (const int32_t*)mpk_find(o->pack, "zapper_i32", &n, 0);
  const double* zf = (const double*)mpk_find(o->pack, "zapper_f64", &n, 0);
  c->zap_cooldown = zi[0]; c->zap_length = zi[1]; c->zap_radius = zi[2];
  c->respawn_frames = zi[3]; c->remove_hit = zi[4];
  c->zap_penalty = zf[0]; c->zap_reward = zf[1];
  c->state_hit_block = (const uint32_t*)mpk_find(o->pack, "state_hit_block", &n, 0);
  mpk_find(o->pack, "resource_cells", &n, 0);
  c->n_res = (int)n;
  size_t sz = (size_t)c->n_res * sizeof(int);
  c->res_piece = (int*)malloc(sz); c->tex_piece = (int*)malloc(sz);
  c->ind_piece = (int*)malloc(sz); c->dmg_piece = (int*)malloc(sz);
  c->health = (int*)malloc(sz); c->frames_since_zapped = (int*)malloc(sz);
  c->active = (int*)malloc(sz); c->claimed_by = (int*)malloc(sz);
  c->destroyed = (int*)malloc(sz);
  return c;
}

void territory_destroy(void* s) {
  Territory* c = (Territory*)s;
  if (!c) return;
  free(c->res_piece); free(c->tex_piece); free(c->ind_piece); free(c->dmg_piece);
  free(c->health); free(c->frames_since_zapped); free(c->active); free(c->claimed_by);
  free(c->destroyed); free(c);
}

int territory_claim_timer(const Oracle* o, int p) { return tr(o)->claim_cooldown[p]; }

/* Extra parity fields of the canonical dump (mirrored by mp_dump):
 *   avat[p][7] = level | freeze << 4 | removal << 12 | noZap << 16 |
 *                movementAllowed << 24 | disallowZapping << 25
 *   glob[3] = claimed resources, glob[5] = sum health, glob[6] = active
 *   resources, glob[7] = sum (claimedBy + 1) */
void territory_dump(const Oracle* o, int32_t* avat, int32_t* glob) {
  const Territory* c = tr(o);
  for (int p = 0; p < o->P; ++p)
    avat[8 * p + 7] = c->level[p] | (o->freeze_counter[p] << 4) |
                      (o->removal_counter[p] << 12) | (c->no_zapping_counter[p] << 16) |
                      (o->movement_allowed[p] << 24) | (c->disallow_zapping[p] << 25);
  int claimed = 0, health = 0, active = 0, by = 0;
  for (int i = 0; i < c->n_res; ++i) {
    if (c->res_piece[i] < 0) continue;   /* not in this episode's map */
    int s = o->pieces[c->res_piece[i]].state;
    claimed += s != c->s_res_unclaimed && s != c->s_res_destroyed;
    health += c->health[i]; active += c->active[i]; by += c->claimed_by[i] + 1;
  }
  glob[3] = claimed; glob[5] = health; glob[6] = active; glob[7] = by;
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
static int res_is_claimed(const Territory* c, int state) {
  return state != c->s_res_unclaimed && state != c->s_res_destroyed;
}

static void tr_start(Oracle* o) {
  Territory* c = tr(o);
  /* by per-kind index (= index into the pack's resource_cells); a resource of a
   * 'choice' map character that is not in this episode's map stays -1 */
  for (int i = 0; i < c->n_res; ++i)
    c->res_piece[i] = c->tex_piece[i] = c->ind_piece[i] = c->dmg_piece[i] = -1;
  for (int i = 0; i < o->npieces; ++i) {
    int idx = o->pieces[i].index;
    switch (o->pieces[i].kind) {
      case MPK_KIND_RESOURCE: c->res_piece[idx] = i; break;
      case MPK_KIND_TEXTURE: c->tex_piece[idx] = i; break;
      case MPK_KIND_REWARD_INDICATOR: c->ind_piece[idx] = i; break;
      case MPK_KIND_DAMAGE_INDICATOR: c->dmg_piece[idx] = i; break;
      case MPK_KIND_MARKING: c->mark_piece[idx] = i; break;
      default: break;
    }
  }
  /* Resource:postStart / RewardIndicator:postStart pair the objects of one
   * cell; they are created together, so equal index == equal cell. */
  for (int i = 0; i < c->n_res; ++i) {
    c->health[i] = 0; c->active[i] = 0; c->claimed_by[i] = -1; c->destroyed[i] = 0;
    c->frames_since_zapped[i] = -1;
    if (c->res_piece[i] < 0) continue;
    if (c->tex_piece[i] < 0 || c->ind_piece[i] < 0 || c->dmg_piece[i] < 0) abort();
    const Piece* r = &o->pieces[c->res_piece[i]];
    const Piece* t = &o->pieces[c->tex_piece[i]];
    if (r->x != t->x || r->y != t->y) abort();
    c->health[i] = c->initial_health; /* Resource:reset */
    c->active[i] = 0; c->claimed_by[i] = -1; c->destroyed[i] = 0;
    c->frames_since_zapped[i] = -1;
  }
  c->ee_t = 1;
  for (int p = 0; p < o->P; ++p) {
    c->level[p] = 1; c->time_since_not_initial[p] = 0;     /* GSM:reset */
    c->disallow_zapping[p] = 0; c->no_zapping_counter[p] = 0; /* Zapper:reset */
    c->claim_cooldown[p] = 0;                                /* ResourceClaimer:reset */
    /* GraduatedSanctionsMarking:postStart (avatar_library.lua:1034-1049): set
     * the state first (a piece without a layer cannot be teleported), then
     * teleport onto the avatar, then connect. */
    const Piece* av = &o->pieces[o->avatar_piece[p]];
    eng_set_state(o, c->mark_piece[p], c->s_mark[c->level[p] - 1]);
    eng_event(o, 8 /* _setLevel -> set_sanctioning_level (:1118) */, p + 1, c->level[p]);
    eng_teleport(o, c->mark_piece[p], av->x, av->y);
    eng_set_orientation(o, c->mark_piece[p], av->orient);
    eng_connect(o, o->avatar_piece[p], c->mark_piece[p]);
  }
}

/* BaseSimulation:update: preUpdate on all, then update on all in creation
 * order: scene, avatars, markings, map objects row-major. */
static void tr_sim_update(Oracle* o) {
  Territory* c = tr(o);
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
  for (int i = 0; i < c->n_res; ++i) {
    if (c->res_piece[i] < 0) continue;
    /* Resource:update (territory/components.lua:193-206) */
    if (c->health[i] < c->initial_health) {
      eng_set_state(o, c->dmg_piece[i], c->s_dmg_damaged);
      if (c->frames_since_zapped[i] >= c->repair_delay) {
        if (eng_u53(o, eng_draw(o, RS_SELF_REPAIR, (uint32_t)i)) < c->thr_repair) {
          c->health[i]++;
          if (c->health[i] == c->initial_health)
            eng_set_state(o, c->dmg_piece[i], c->s_dmg_inactive);
        }
      }
      c->frames_since_zapped[i]++;
    }
    /* RewardIndicator:update (:299-308): "dry_" .. resourceState */
    int rs = o->pieces[c->res_piece[i]].state, owner = -1;
    for (int p = 0; p < o->P; ++p) if (rs == c->s_claimed[p]) owner = p;
    if (c->active[i] && owner >= 0) eng_set_state(o, c->ind_piece[i], c->s_dry[owner]);
    else eng_set_state(o, c->ind_piece[i], c->s_ind_inactive);
  }
}

static void tr_run_updaters(Oracle* o) {
  Territory* c = tr(o);
  int order[ORC_MAX_PLAYERS];
  const int P = o->P;
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
  /* 135: Zapper respawn (framesTillRespawn = 1e6: effectively never) */
  eng_trace(o, 135, "Zapper.respawn");
  for (int p = 0; p < P; ++p) {
    int piece = o->avatar_piece[p];
    if (is_wait(o, p) && eng_frames(o, piece) >= c->respawn_frames) abort();
  }
  /* 130: Paintbrush drawBrush (territory/components.lua:401-411): every frame */
  eng_trace(o, 130, "Paintbrush.drawBrush");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_BRUSH, order, P);
  for (int i = 0; i < P; ++i)
    eng_hit_beam(o, o->avatar_piece[order[i]], c->hit_brush[order[i]], 1, 0);
  /* 100: StochasticIntervalEpisodeEnding */
  eng_trace(o, 100, "StochasticIntervalEpisodeEnding.maybeEndEpisode");
  if (eng_frames(o, 0) >= c->ee_min_frames && c->ee_t % c->ee_interval == 0)
    if (eng_u53(o, eng_draw(o, RS_EPISODE_END, 0)) < c->thr_ee) o->continue_flag = 0;
  /* 100: ResourceClaimer claim (territory/components.lua:255-275) */
  eng_trace(o, 100, "ResourceClaimer.claim");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_CLAIM, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i];
    if (c->claim_wait < 0) continue;
    if (c->claim_cooldown[p] > 0) c->claim_cooldown[p]--;
    else if (o->action[p][ACT_FIRE_CLAIM] == 1) {
      c->claim_cooldown[p] = c->claim_wait;
      eng_hit_beam(o, o->avatar_piece[p], c->hit_claim[p], c->claim_length, c->claim_radius);
    }
  }
  /* 100: Resource provideRewards: group claimedResources, probability
   * rewardRate, startFrame rewardDelay (territory/components.lua:85-102). */
  eng_trace(o, 100, "Resource.provideRewards");
  for (int i = 0; i < c->n_res; ++i) {
    int piece = c->res_piece[i];
    if (piece < 0 || !res_is_claimed(c, o->pieces[piece].state)) continue;
    if (eng_frames(o, piece) < c->reward_delay) continue;
    if (eng_u53(o, eng_draw(o, RS_RESOURCE_REWARD, (uint32_t)i)) >= c->thr_reward) continue;
    if (c->claimed_by[i] >= 0) add_reward(o, c->claimed_by[i], c->reward); /* Taste 'none' */
    c->active[i] = 1;
  }
  /* 3: GraduatedSanctionsMarking resetToInitialLevel (avatar_library.lua:1010-1026) */
  eng_trace(o, 3, "GraduatedSanctionsMarking.resetToInitialLevel");
  for (int p = 0; p < P; ++p) {
    if (c->level[p] != 1 && is_alive(o, p)) {
      c->time_since_not_initial[p]++;
      if (c->time_since_not_initial[p] == c->recovery_time) {
        c->level[p] = 1;
        eng_set_state(o, c->mark_piece[p], c->s_mark[0]);
        eng_event(o, 8 /* _setLevel */, p + 1, 1);
        c->time_since_not_initial[p] = 0;
      }
    }
  }
  /* 2: Resource releaseClaimOfDeadAgent: startFrame 5 (:103-117) */
  eng_trace(o, 2, "Resource.releaseClaimOfDeadAgent");
  for (int i = 0; i < c->n_res; ++i) {
    int piece = c->res_piece[i];
    if (piece < 0 || !res_is_claimed(c, o->pieces[piece].state)) continue;
    if (eng_frames(o, piece) < 5) continue;
    if (c->claimed_by[i] >= 0 && is_wait(o, c->claimed_by[i]) && !c->destroyed[i]) {
      eng_set_state(o, piece, c->s_res_unclaimed);
      c->active[i] = 0;
      c->claimed_by[i] = -1;
    }
  }
}

/* Resource:_claim (territory/components.lua:119-137) */
static void claim(Oracle* o, Territory* c, int i, int player) {
  c->claimed_by[i] = player;
  int piece = c->res_piece[i];
  if (o->pieces[piece].state != c->s_claimed[player] && !c->destroyed[i]) {
    eng_set_state(o, piece, c->s_claimed[player]);
    c->active[i] = 0;
    eng_event(o, 4 /* claimed_resource (territory/components.lua:133) */, player + 1, 0);
  }
}

static int tr_on_hit(Oracle* o, int target, int hitter, int hit) {
  Territory* c = tr(o);
  const Piece* t = &o->pieces[target];
  int blocked = 0;
  if (c->state_hit_block[t->state] & (1u << hit)) blocked = 1; /* AllBeamBlocker */
  int hp = o->pieces[hitter].index; /* hitting avatar */
  if (t->kind == MPK_KIND_AVATAR && hit == c->hit_zap) {
    /* Zapper:onHit (avatar_library.lua:652-681), removeHitPlayer = false */
    eng_event(o, 1 /* zap (avatar_library.lua:661) */, hp + 1, t->index + 1);
    add_reward(o, t->index, c->zap_penalty);
    add_reward(o, hp, c->zap_reward);
    if (c->remove_hit) eng_set_state(o, target, o->wait_state[t->index]);
    blocked = 1;
  } else if (t->kind == MPK_KIND_MARKING && hit == c->hit_zap) {
    /* GraduatedSanctionsMarking:onHit (avatar_library.lua:1051-1097) */
    int p = t->index, l = c->level[p] - 1;
    add_reward(o, hp, c->lv_source[l]);
    add_reward(o, p, c->lv_target[l]);
    c->level[p] += c->lv_increment[l];
    if (c->lv_remove[l]) {
      o->removal_counter[p] = 1;                           /* removeAfterDelay(1) */
      o->movement_allowed[p] = 0; o->freeze_counter[p] = 1; /* disallowMovementUntil(1) */
      c->disallow_zapping[p] = 1; c->no_zapping_counter[p] = 1;
      eng_event(o, 7 /* removal_due_to_sanctioning (:1070) */, hp + 1, p + 1);
    } else {
      eng_set_state(o, target, c->s_mark[c->level[p] - 1]); /* _setLevel */
      eng_event(o, 8 /* set_sanctioning_level (:1118) */, p + 1, c->level[p]);
      if (c->lv_freeze[l] > 0) {
        o->movement_allowed[p] = 0; o->freeze_counter[p] = c->lv_freeze[l];
        c->disallow_zapping[p] = 1; c->no_zapping_counter[p] = c->lv_freeze[l];
      }
    }
    c->time_since_not_initial[p] = 0;
    eng_event(o, 6 /* sanctioning (:1088) */, hp + 1, p + 1);
  } else if (t->kind == MPK_KIND_RESOURCE) {
    /* Resource:onHit (territory/components.lua:139-185) */
    int i = t->index;
    for (int p = 0; p < o->P; ++p) {
      if (hit == c->hit_brush[p]) claim(o, c, i, hp);
      if (hit == c->hit_claim[p]) { claim(o, c, i, hp); return blocked; } /* passes through */
    }
    if (hit == c->hit_zap) {
      c->health[i]--;
      c->frames_since_zapped[i] = 0;
      if (c->health[i] == 0) {
        c->health[i] = c->initial_health;
        eng_set_state(o, target, c->s_res_destroyed);
        c->active[i] = 0;
        eng_set_state(o, c->tex_piece[i], c->s_tex_destroyed);
        eng_set_state(o, c->dmg_piece[i], c->s_dmg_inactive);
        c->destroyed[i] = 1;
        eng_event(o, 5 /* destroyed_resource (territory/components.lua:168) */, hp + 1, 0);
        return blocked; /* zaps pass through a destroyed resource */
      }
      blocked = 1;
    }
  }
  return blocked;
}

static void tr_on_enter(Oracle* o, int target, int entering, int contact) {
  (void)o; (void)target; (void)entering; (void)contact; /* nothing reacts to contact */
}

static void tr_on_state_change(Oracle* o, int piece, int old_state) {
  Territory* c = tr(o);
  const Piece* p = &o->pieces[piece];
  if (p->kind != MPK_KIND_AVATAR) return;
  int pl = p->index; /* Avatar:onStateChange (avatar_library.lua:430-453) */
  if (old_state == o->wait_state[pl] && p->state == o->alive_state[pl]) {
    o->freeze_counter[pl] = 0; o->removal_counter[pl] = 0;
  } else if (old_state == o->alive_state[pl] && p->state == o->wait_state[pl]) {
    /* 'die' -> GraduatedSanctionsMarking:avatarStateChange */
    eng_set_state(o, c->mark_piece[pl], c->s_mark_wait);
  }
}

const SubstrateVtbl kTerritoryVtbl = {
    tr_on_enter, tr_on_hit, tr_on_state_change,
    tr_sim_update, tr_run_updaters, tr_start,
};
