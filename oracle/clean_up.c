/* ORACLE — test infrastructure only.  See engine.h.
 *
 * clean_up rules: restatement of the reference's Lua components
 *   lua/levels/clean_up/components.lua  (AppleGrow, DirtTracker, DirtCleaning,
 *     Cleaner, RiverMonitor, DirtSpawner, Edible, Taste, GlobalData,
 *     AllNonselfCumulants)
 *   lua/modules/avatar_library.lua      (Avatar, Zapper, ReadyToShootObservation)
 *   lua/modules/component_library.lua   (Animation,
 *     StochasticIntervalEpisodeEnding, BeamBlocker)
 * with kwargs from configs/substrates/clean_up.py (carried by the pack).
 */
#include <stdlib.h>
#include <string.h>

#include "../include/mp_pack.h"
#include "engine.h"

enum { HIT_ZAP = 0, HIT_CLEAN = 1 };
enum { ACT_MOVE = 0, ACT_TURNA = 1, ACT_FIRE_ZAP = 2, ACT_FIRE_CLEAN = 3 };

typedef struct {
  int n_apple, n_dirt, n_water;
  int *apple_piece, *dirt_piece, *water_piece;
  /* state ids */
  int s_apple, s_apple_wait, s_dirt, s_dirt_wait, s_water[4];
  const uint32_t* state_hit_block;
  /* component kwargs */
  int zap_cooldown, zap_length, zap_radius, respawn_frames, remove_hit;
  double zap_penalty, zap_reward;
  int clean_cooldown, clean_length, clean_radius;
  int dirt_delay, ee_min_frames, ee_interval, anim_frames;
  double max_growth, thr_depletion, thr_restoration, dirt_prob, ee_prob,
      eat_reward;
  /* RiverMonitor (clean_up/components.lua:262-298) */
  int dirt_count, clean_count;
  /* DirtSpawner (:300-348): potential set as a flag per dirt site (site order
   * == piece id order == set.toSortedList order) */
  uint8_t* potential;
  int time_step;
  /* StochasticIntervalEpisodeEnding._t (component_library.lua:942-948) */
  int ee_t;
  /* Cleaner / Taste / GlobalData / AllNonselfCumulants volatile variables */
  int clean_timer[ORC_MAX_PLAYERS];
  int player_cleaned[ORC_MAX_PLAYERS], player_ate[ORC_MAX_PLAYERS];
  int cleaned_this_step[ORC_MAX_PLAYERS], ate_this_step[ORC_MAX_PLAYERS];
  double others_cleaned[ORC_MAX_PLAYERS], others_ate[ORC_MAX_PLAYERS];
} CleanUp;

static CleanUp* cu(const Oracle* o) { return (CleanUp*)o->sub_state; }

void* clean_up_create(Oracle* o) {
  CleanUp* c = (CleanUp*)calloc(1, sizeof(CleanUp));
  uint64_t n;
  const int32_t* st = (const int32_t*)mpk_find(o->pack, "cu_states", &n, 0);
  c->s_apple = st[0]; c->s_apple_wait = st[1];
  c->s_dirt = st[2]; c->s_dirt_wait = st[3];
  for (int i = 0; i < 4; ++i) c->s_water[i] = st[4 + i];
  const int32_t* zi = (const int32_t*)mpk_find(o->pack, "zapper_i32", &n, 0);
  c->zap_cooldown = zi[0]; c->zap_length = zi[1]; c->zap_radius = zi[2];
  c->respawn_frames = zi[3]; c->remove_hit = zi[4];
  const double* zf = (const double*)mpk_find(o->pack, "zapper_f64", &n, 0);
  c->zap_penalty = zf[0]; c->zap_reward = zf[1];
  const int32_t* ci = (const int32_t*)mpk_find(o->pack, "cu_i32", &n, 0);
  c->clean_cooldown = ci[0]; c->clean_length = ci[1]; c->clean_radius = ci[2];
  c->dirt_delay = ci[3]; c->ee_min_frames = ci[4]; c->ee_interval = ci[5];
  c->anim_frames = ci[6];
  const double* cf = (const double*)mpk_find(o->pack, "cu_f64", &n, 0);
  c->max_growth = cf[0]; c->thr_depletion = cf[1]; c->thr_restoration = cf[2];
  c->dirt_prob = cf[3]; c->ee_prob = cf[4]; c->eat_reward = cf[5];
  c->state_hit_block =
      (const uint32_t*)mpk_find(o->pack, "state_hit_block", &n, 0);
  mpk_find(o->pack, "apple_cells", &n, 0); c->n_apple = (int)n;
  mpk_find(o->pack, "dirt_cells", &n, 0); c->n_dirt = (int)n;
  mpk_find(o->pack, "water_cells", &n, 0); c->n_water = (int)n;
  c->apple_piece = (int*)calloc((size_t)c->n_apple, sizeof(int));
  c->dirt_piece = (int*)calloc((size_t)c->n_dirt, sizeof(int));
  c->water_piece = (int*)calloc((size_t)c->n_water, sizeof(int));
  c->potential = (uint8_t*)calloc((size_t)c->n_dirt, 1);
  return c;
}

void clean_up_destroy(void* s) {
  CleanUp* c = (CleanUp*)s;
  if (!c) return;
  free(c->apple_piece); free(c->dirt_piece); free(c->water_piece);
  free(c->potential); free(c);
}

static int player_of(const Oracle* o, int piece) { return o->pieces[piece].index; }

static int is_alive(const Oracle* o, int p) {
  /* Avatar:isAlive (avatar_library.lua:491-493) */
  return o->pieces[o->avatar_piece[p]].state == o->alive_state[p];
}

/* Avatar:addReward with skipWaitStateRewards (avatar_library.lua:362-376) */
static void add_reward(Oracle* o, int p, double amount) {
  if (o->pieces[o->avatar_piece[p]].state != o->wait_state[p])
    o->reward[p] += amount;
}

/* reset() on all, then postStart() (base_simulation.lua:450-471,441-444). */
static void cu_start(Oracle* o) {
  CleanUp* c = cu(o);
  int na = 0, nd = 0, nw = 0;
  for (int i = 0; i < o->npieces; ++i) {
    Piece* p = &o->pieces[i];
    if (p->kind == MPK_KIND_APPLE_GROW) c->apple_piece[na++] = i;
    else if (p->kind == MPK_KIND_DIRT) c->dirt_piece[nd++] = i;
    else if (p->kind == MPK_KIND_ANIM) c->water_piece[nw++] = i;
  }
  /* RiverMonitor:reset, DirtSpawner:reset, episode-ending reset */
  c->dirt_count = c->clean_count = 0;
  memset(c->potential, 0, (size_t)c->n_dirt);
  c->time_step = 1;
  c->ee_t = 1;
  for (int p = 0; p < o->P; ++p) {
    c->clean_timer[p] = 0; /* Cleaner:reset (:234-237) */
    c->player_cleaned[p] = c->player_ate[p] = 0;
    c->cleaned_this_step[p] = c->ate_this_step[p] = 0; /* GlobalData:reset */
    c->others_cleaned[p] = c->others_ate[p] = 0.0;
  }
  /* DirtTracker:postStart (:103-116) */
  for (int i = 0; i < c->n_dirt; ++i) {
    int s = o->pieces[c->dirt_piece[i]].state;
    if (s == c->s_dirt_wait) { c->potential[i] = 1; c->clean_count++; }
    else if (s == c->s_dirt) c->dirt_count++;
  }
  /* Animation:postStart with randomStartFrame (component_library.lua:1064) */
  for (int i = 0; i < c->n_water; ++i) {
    uint32_t k = eng_bounded(o, eng_draw(o, RS_ANIM_START, (uint32_t)i), 4u);
    eng_set_state(o, c->water_piece[i], c->s_water[k]);
  }
}

/* BaseSimulation:update (base_simulation.lua:476-486): preUpdate on all, then
 * update on all, objects in creation order: scene, avatars, map objects. */
static void cu_sim_update(Oracle* o) {
  CleanUp* c = cu(o);
  for (int p = 0; p < o->P; ++p) o->reward[p] = 0.0; /* Avatar:preUpdate :330 */

  /* scene: DirtSpawner:update (clean_up/components.lua:329-340) */
  if (c->time_step > c->dirt_delay) {
    PhiloxOut d = eng_draw(o, RS_DIRT_SPAWN, 0);
    double u = (double)eng_u53(o, d) * (1.0 / 9007199254740992.0);
    if (u < c->dirt_prob) {
      int n = 0;
      for (int i = 0; i < c->n_dirt; ++i) n += c->potential[i];
      if (n > 0) { /* random:choice(set.toSortedList(potential)) */
        int k = (int)eng_bounded(o, d, (uint32_t)n);
        for (int i = 0; i < c->n_dirt; ++i)
          if (c->potential[i] && k-- == 0) {
            eng_set_state(o, c->dirt_piece[i], c->s_dirt);
            break;
          }
      }
    }
  }
  c->time_step++;
  c->ee_t++; /* StochasticIntervalEpisodeEnding:update */

  /* avatars: Avatar:update (avatar_library.lua:334-355), Zapper:update */
  for (int p = 0; p < o->P; ++p) {
    if (o->freeze_counter[p] == 1) o->movement_allowed[p] = 1;
    if (o->freeze_counter[p] > 0) o->freeze_counter[p]--;
    if (o->removal_counter[p] == 1)
      eng_set_state(o, o->avatar_piece[p], o->wait_state[p]);
    if (o->removal_counter[p] > 0) o->removal_counter[p]--;
  }

  /* potential apples: AppleGrow:update (clean_up/components.lua:64-80) */
  double dirt_fraction =
      (double)c->dirt_count / (double)(c->dirt_count + c->clean_count);
  double interpolation = (dirt_fraction - c->thr_depletion) /
                         (c->thr_restoration - c->thr_depletion);
  if (interpolation > 1.0) interpolation = 1.0;
  double probability = c->max_growth * interpolation;
  for (int i = 0; i < c->n_apple; ++i) {
    double u = (double)eng_u53(o, eng_draw(o, RS_APPLE_GROW, (uint32_t)i)) *
               (1.0 / 9007199254740992.0);
    if (u < probability) eng_set_state(o, c->apple_piece[i], c->s_apple);
  }
}

/* Updaters, priority descending (updater_registry.lua:166-173,260-273).
 * Same-priority order is unspecified in the reference (pairs()); fixed here as
 * written. */
static void cu_run_updaters(Oracle* o) {
  CleanUp* c = cu(o);
  int order[ORC_MAX_PLAYERS];
  const int P = o->P;

  /* 400: Cleaner / Taste / AllNonselfCumulants resets
   * (clean_up/components.lua:226-232,427-434,547-556) */
  eng_trace(o, 400, "Cleaner.resetCumulant");
  eng_trace(o, 400, "Taste.resetCumulant");
  eng_trace(o, 400, "AllNonselfCumulants.resetCumulants");
  for (int p = 0; p < P; ++p) {
    c->player_cleaned[p] = 0; c->player_ate[p] = 0;
    c->others_cleaned[p] = 0.0; c->others_ate[p] = 0.0;
  }

  /* 150: Avatar move (avatar_library.lua:155-203), probability = speed = 1 */
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

  /* 140: Cleaner clean (clean_up/components.lua:201-224) */
  eng_trace(o, 140, "Cleaner.clean");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_CLEAN, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i];
    if (!is_alive(o, p) || c->clean_cooldown < 0) continue;
    if (c->clean_timer[p] > 0) c->clean_timer[p]--;
    else if (o->action[p][ACT_FIRE_CLEAN] == 1) {
      c->clean_timer[p] = c->clean_cooldown;
      eng_hit_beam(o, o->avatar_piece[p], HIT_CLEAN, c->clean_length,
                   c->clean_radius);
    }
  }

  /* 135: Zapper respawn, state = waitState, startFrame = framesTillRespawn
   * (avatar_library.lua:638-649) */
  eng_trace(o, 135, "Zapper.respawn");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_RESPAWN, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i], piece = o->avatar_piece[p];
    if (o->pieces[piece].state != o->wait_state[p]) continue;
    if (eng_frames(o, piece) < c->respawn_frames) continue;
    eng_teleport_to_group(o, piece, (uint32_t)o->spawn_group_mask,
                          o->alive_state[p], TELEPORT_PICK_RANDOM, RS_RESPAWN,
                          p);
  }

  /* 100: StochasticIntervalEpisodeEnding (component_library.lua:927-940),
   * startFrame = minimumFramesPerEpisode on the scene piece (piece 0). */
  eng_trace(o, 100, "StochasticIntervalEpisodeEnding.maybeEndEpisode");
  if (eng_frames(o, 0) >= c->ee_min_frames && c->ee_t % c->ee_interval == 0) {
    double u = (double)eng_u53(o, eng_draw(o, RS_EPISODE_END, 0)) *
               (1.0 / 9007199254740992.0);
    if (u < c->ee_prob) o->continue_flag = 0; /* simulation:endEpisode() */
  }

  /* 100: Animation (component_library.lua:1070-1094): state k -> k+1 after
   * gameFramesPerAnimationFrame frames in state, looping. */
  eng_trace(o, 100, "Animation");
  for (int i = 0; i < c->n_water; ++i) {
    int piece = c->water_piece[i];
    if (eng_frames(o, piece) < c->anim_frames) continue;
    for (int k = 0; k < 4; ++k)
      if (o->pieces[piece].state == c->s_water[k]) {
        eng_set_state(o, piece, c->s_water[(k + 1) & 3]);
        break;
      }
  }
  /* 4: AllNonselfCumulants.getCumulants (clean_up/components.lua:535-545) */
  eng_trace(o, 4, "AllNonselfCumulants.getCumulants");
  for (int p = 0; p < P; ++p) {
    int sc = 0, sa = 0;
    for (int q = 0; q < P; ++q)
      if (q != p) { sc += c->cleaned_this_step[q]; sa += c->ate_this_step[q]; }
    c->others_cleaned[p] = (double)sc;
    c->others_ate[p] = (double)sa;
  }
  /* 2: GlobalData.resetCumulants (:483-492) */
  eng_trace(o, 2, "GlobalData.resetCumulants");
  for (int p = 0; p < P; ++p) c->cleaned_this_step[p] = c->ate_this_step[p] = 0;
}

static int cu_on_hit(Oracle* o, int target, int hitter, int hit) {
  CleanUp* c = cu(o);
  const Piece* t = &o->pieces[target];
  int blocked = 0;
  /* BeamBlocker:onHit (component_library.lua:678-685) */
  if (c->state_hit_block[t->state] & (1u << hit)) blocked = 1;
  if (t->kind == MPK_KIND_AVATAR && hit == HIT_ZAP) {
    /* Zapper:onHit (avatar_library.lua:652-681) */
    int zapped = player_of(o, target), zapper = player_of(o, hitter);
    eng_event(o, 1 /* zap */, zapper + 1, zapped + 1);
    o->zap_matrix[zapped][zapper]++; o->num_zapped[zapper]++;
    add_reward(o, zapped, c->zap_penalty);
    add_reward(o, zapper, c->zap_reward);
    if (c->remove_hit) eng_set_state(o, target, o->wait_state[zapped]);
    blocked = 1;
  }
  if (t->kind == MPK_KIND_DIRT && hit == HIT_CLEAN && t->state == c->s_dirt) {
    /* DirtCleaning:onHit (clean_up/components.lua:141-157) */
    eng_set_state(o, target, c->s_dirt_wait);
    int p = player_of(o, hitter);
    eng_event(o, 3 /* player_cleaned (:152) */, p + 1, 0);
    /* Taste:cleaned with role 'free': no reward (:436-444) */
    c->player_cleaned[p]++;        /* Cleaner:setCumulant (:247-255) */
    c->cleaned_this_step[p] = 1;   /* GlobalData:setCleanedThisStep */
    blocked = 1;
  }
  return blocked;
}

static void cu_on_enter(Oracle* o, int target, int entering, int contact) {
  CleanUp* c = cu(o);
  (void)contact; /* the only contact in clean_up is 'avatar' */
  const Piece* t = &o->pieces[target];
  /* Edible:onEnter (clean_up/components.lua:390-408) */
  if (t->kind == MPK_KIND_APPLE_GROW && t->state == c->s_apple) {
    int p = player_of(o, entering);
    add_reward(o, p, c->eat_reward); /* Taste:consumed, role 'free' (:446-455) */
    eng_event(o, 2 /* edible_consumed (:402) */, p + 1, 0);
    c->player_ate[p]++;
    c->ate_this_step[p] = 1;
    eng_set_state(o, target, c->s_apple_wait);
  }
}

static void cu_on_state_change(Oracle* o, int piece, int old_state) {
  CleanUp* c = cu(o);
  const Piece* p = &o->pieces[piece];
  if (p->kind == MPK_KIND_DIRT) {
    /* DirtTracker:onStateChange (clean_up/components.lua:118-129) */
    int site = p->index;
    if (old_state == c->s_dirt_wait && p->state == c->s_dirt) {
      c->dirt_count++; c->clean_count--; c->potential[site] = 0;
    } else if (old_state == c->s_dirt && p->state == c->s_dirt_wait) {
      c->dirt_count--; c->clean_count++; c->potential[site] = 1;
    }
  } else if (p->kind == MPK_KIND_AVATAR) {
    /* Avatar:onStateChange (avatar_library.lua:430-453) */
    int pl = p->index;
    if (old_state == o->wait_state[pl] && p->state == o->alive_state[pl]) {
      o->freeze_counter[pl] = 0;
      o->removal_counter[pl] = 0;
    }
  }
}

const SubstrateVtbl kCleanUpVtbl = {
    cu_on_enter, cu_on_hit, cu_on_state_change,
    cu_sim_update, cu_run_updaters, cu_start,
};

double clean_up_num_others_cleaned(const Oracle* o, int player) {
  return cu(o)->others_cleaned[player];
}
int clean_up_clean_timer(const Oracle* o, int player) {
  return cu(o)->clean_timer[player];
}
/* which: 0 Cleaner.player_cleaned, 1 Taste.player_ate_apple,
 * 2 AllNonselfCumulants.num_others_who_ate_this_step */
double clean_up_debug_metric(const Oracle* o, int player, int which) {
  const CleanUp* c = cu(o);
  return which == 0 ? (double)c->player_cleaned[player]
       : which == 1 ? (double)c->player_ate[player] : c->others_ate[player];
}
int clean_up_dirt_count(const Oracle* o) { return cu(o)->dirt_count; }
