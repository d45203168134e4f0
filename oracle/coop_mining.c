/* ORACLE — test infrastructure only.  See engine.h.
 *
 * coop_mining rules: restatement of the reference's Lua components
 *   lua/levels/coop_mining/components.lua  (FixedRateRegrow :29-60, Ore :62-143,
 *                                           MineBeam :147-254, MiningTracker :256-283)
 *   lua/modules/component_library.lua:907-948  (StochasticIntervalEpisodeEnding),
 *                                    :667-685  (BeamBlocker)
 *   lua/modules/avatar_library.lua             (Avatar; no Zapper in this level)
 * with kwargs from configs/substrates/coop_mining.py (in the pack).
 *
 * One ore OBJECT carries one `Ore` component per ore type over one state machine
 * (oreWait / <type>Raw / <type>Partial); each component keeps its own miners and its own
 * countdown.  Type k (0-based) is the component with minNumMiners == k + 1: the Lua hands
 * minNumMiners to the reward tables as the ore-type index (Ore:onHit, :124,129).
 */
#include <stdlib.h>
#include <string.h>

#include "../include/mp_pack.h"
#include "engine.h"

enum { ACT_MOVE = 0, ACT_TURNA = 1, ACT_MINE = 2 };
enum { CM_TYPES = 2 };

typedef struct {
  int n_ore;
  int* ore_piece;                 /* ore pieces in creation order */
  /* Ore._miners / Ore._miningCountdown, per ore object and per component (type) */
  uint32_t* miners;               /* [n_ore][CM_TYPES] bit p = player p is in _miners */
  int* countdown;                 /* [n_ore][CM_TYPES] */
  int s_wait, s_raw[CM_TYPES], s_partial[CM_TYPES];
  int min_miners[CM_TYPES], window[CM_TYPES];
  int cooldown, beam_length, beam_radius, hit_mine;
  const double* reward;           /* [P][2 * CM_TYPES]: mining per type, extracting per type */
  const uint64_t* thr;            /* regrow per type, then episode end */
  const uint32_t* state_hit_block;
  int ee_min_frames, ee_interval, ee_t;
} Coop;

static Coop* cm(const Oracle* o) { return (Coop*)o->sub_state; }

void* coop_create(Oracle* o) {
  Coop* c = (Coop*)calloc(1, sizeof(Coop));
  uint64_t n;
  const int32_t* st = (const int32_t*)mpk_find(o->pack, "cm_states", &n, 0);
  const int32_t* ci = (const int32_t*)mpk_find(o->pack, "cm_i32", &n, 0);
  c->s_wait = st[0];
  for (int k = 0; k < CM_TYPES; ++k) { c->s_raw[k] = st[1 + k]; c->s_partial[k] = st[1 + CM_TYPES + k]; }
  c->cooldown = ci[0]; c->beam_length = ci[1]; c->beam_radius = ci[2]; c->hit_mine = ci[3];
  c->ee_min_frames = ci[4]; c->ee_interval = ci[5];
  for (int k = 0; k < CM_TYPES; ++k) { c->min_miners[k] = ci[6 + 2 * k]; c->window[k] = ci[7 + 2 * k]; }
  c->reward = (const double*)mpk_find(o->pack, "cm_f64", &n, 0);
  c->thr = (const uint64_t*)mpk_find(o->pack, "cm_thr", &n, 0);
  c->state_hit_block = (const uint32_t*)mpk_find(o->pack, "state_hit_block", &n, 0);
  mpk_find(o->pack, "ore_cells", &n, 0);
  c->n_ore = (int)n;
  c->ore_piece = (int*)calloc((size_t)c->n_ore, sizeof(int));
  c->miners = (uint32_t*)calloc((size_t)c->n_ore * CM_TYPES, sizeof(uint32_t));
  c->countdown = (int*)calloc((size_t)c->n_ore * CM_TYPES, sizeof(int));
  return c;
}

void coop_destroy(void* s) {
  Coop* c = (Coop*)s;
  if (!c) return;
  free(c->ore_piece); free(c->miners); free(c->countdown); free(c);
}

/* what the state dump carries of the Lua-side variables: ores not waiting; the sum of the
 * live countdowns; a position-weighted sum of the miner sets */
void coop_dump(const Oracle* o, int32_t* glob) {
  const Coop* c = cm(o);
  int live = 0;
  uint32_t cd = 0, ms = 0;
  for (int i = 0; i < c->n_ore; ++i) {
    live += o->pieces[c->ore_piece[i]].state != c->s_wait;
    for (int k = 0; k < CM_TYPES; ++k) {
      const int v = c->countdown[i * CM_TYPES + k];
      cd += (uint32_t)(v > 0 ? v : 0);
      ms += c->miners[i * CM_TYPES + k] * (uint32_t)(i + 1);
    }
  }
  glob[3] = live; glob[5] = (int32_t)cd; glob[6] = (int32_t)(ms & 0x7fffffffu);
}

int coop_cooldown(const Oracle* o) { return cm(o)->cooldown; }

static void add_reward(Oracle* o, int p, double amount) {
  /* Avatar:addReward with skipWaitStateRewards (avatar_library.lua:362-376) */
  if (o->pieces[o->avatar_piece[p]].state != o->wait_state[p]) o->reward[p] += amount;
}

/* Ore:reset (components.lua:90-97) */
static void ore_reset(Oracle* o, int i, int k) {
  Coop* c = cm(o);
  c->miners[i * CM_TYPES + k] = 0;
  c->countdown[i * CM_TYPES + k] = 0;
  if (o->pieces[c->ore_piece[i]].state != c->s_wait) eng_set_state(o, c->ore_piece[i], c->s_raw[k]);
}

static void cm_start(Oracle* o) {
  Coop* c = cm(o);
  int n = 0;
  for (int i = 0; i < o->npieces; ++i)
    if (o->pieces[i].kind == MPK_KIND_ORE) c->ore_piece[n++] = i;
  c->ee_t = 1;
  /* Ore:reset on every component (the objects start in oreWait: no setState) */
  for (int i = 0; i < c->n_ore; ++i)
    for (int k = 0; k < CM_TYPES; ++k) ore_reset(o, i, k);
  /* MineBeam:start: _coolingTimer = 0 (zap_timer, cleared by the episode start) */
}

/* BaseSimulation:update (base_simulation.lua:476-486): preUpdate on all, then update on
 * all, objects in creation order: scene, avatars, map objects. */
static void cm_sim_update(Oracle* o) {
  Coop* c = cm(o);
  for (int p = 0; p < o->P; ++p) o->reward[p] = 0.0; /* Avatar:preUpdate; MiningTracker:preUpdate */
  c->ee_t++; /* StochasticIntervalEpisodeEnding:update */
  for (int p = 0; p < o->P; ++p) {
    /* Avatar:update (avatar_library.lua:334-355) */
    if (o->freeze_counter[p] == 1) o->movement_allowed[p] = 1;
    if (o->freeze_counter[p] > 0) o->freeze_counter[p]--;
    if (o->removal_counter[p] == 1) eng_set_state(o, o->avatar_piece[p], o->wait_state[p]);
    if (o->removal_counter[p] > 0) o->removal_counter[p]--;
    /* MineBeam:update (components.lua:228-244): the timer runs down FIRST, and a beam
     * leaves in the very update that brings it to zero */
    if (o->zap_timer[p] > 0) o->zap_timer[p]--;
    if (o->action[p][ACT_MINE] == 1 && o->zap_timer[p] == 0) {   /* readyToShoot() >= 1 */
      o->zap_timer[p] = c->cooldown;
      eng_hit_beam(o, o->avatar_piece[p], c->hit_mine, c->beam_length, c->beam_radius);
    }
  }
  /* Ore:update, component by component (components.lua:99-105) */
  for (int i = 0; i < c->n_ore; ++i)
    for (int k = 0; k < CM_TYPES; ++k) {
      int* cd = &c->countdown[i * CM_TYPES + k];
      *cd -= 1;
      if (*cd == 0) ore_reset(o, i, k);
      if (*cd < -1) *cd = -1;   /* (Lua counts down for ever; only 0 matters) */
    }
}

static void cm_run_updaters(Oracle* o) {
  Coop* c = cm(o);
  int order[ORC_MAX_PLAYERS];
  const int P = o->P;
  /* 200: FixedRateRegrow, one engine-side probabilistic updater per live state on the
   * pieces in waitState (components.lua:45-60).  A12: every piece of the group is selected
   * independently, one draw per piece and updater; A11: registration order. */
  eng_trace(o, 200, "FixedRateRegrow.regrow");
  const int upper = o->avatar_layer;   /* 'upperPhysical': where the avatars stand */
  for (int k = 0; k < CM_TYPES; ++k)
    for (int i = 0; i < c->n_ore; ++i) {
      const Piece* pc = &o->pieces[c->ore_piece[i]];
      if (pc->state != c->s_wait) continue;
      if (eng_u53(o, eng_draw(o, RS_REGROW, (uint32_t)(k * c->n_ore + i))) >= c->thr[k]) continue;
      if (eng_cell(o, upper, pc->x, pc->y) >= 0) continue;   /* queryPosition('upperPhysical') */
      eng_set_state(o, c->ore_piece[i], c->s_raw[k]);
    }
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
  /* 100: StochasticIntervalEpisodeEnding (component_library.lua:927-940) */
  eng_trace(o, 100, "StochasticIntervalEpisodeEnding.maybeEndEpisode");
  if (eng_frames(o, 0) >= c->ee_min_frames && c->ee_t % c->ee_interval == 0) {
    if (eng_u53(o, eng_draw(o, RS_EPISODE_END, 0)) < c->thr[CM_TYPES]) o->continue_flag = 0;
  }
}

/* Ore:onHit of component k (components.lua:113-143) */
static int ore_on_hit(Oracle* o, int i, int k, int hitter_player) {
  Coop* c = cm(o);
  const int piece = c->ore_piece[i];
  const int state = o->pieces[piece].state;
  if (state != c->s_raw[k] && state != c->s_partial[k]) return 0;
  /* Ore:addMiner */
  c->countdown[i * CM_TYPES + k] = c->window[k];
  c->miners[i * CM_TYPES + k] |= 1u << hitter_player;
  eng_set_state(o, piece, c->s_partial[k]);
  /* MineBeam:processRoleMineEvent(minNumMiners): the type index IS minNumMiners */
  add_reward(o, hitter_player, c->reward[hitter_player * 2 * CM_TYPES + k]);
  eng_event(o, 13 /* mining (components.lua:196) */, hitter_player + 1, k + 1);
  const uint32_t m = c->miners[i * CM_TYPES + k];
  if (__builtin_popcount(m) == c->min_miners[k]) {
    for (int id = 0; id < o->P; ++id) {
      if (!((m >> id) & 1u)) continue;
      /* MineBeam:processRoleExtractEvent / processRolePairExtractEvent */
      add_reward(o, id, c->reward[id * 2 * CM_TYPES + CM_TYPES + k]);
      eng_event(o, 14 /* extraction (:210) */, id + 1, k + 1);
      for (int other = 0; other < o->P; ++other)
        if (other != id && ((m >> other) & 1u))
          eng_event(o, 15 /* extraction_pair (:220) */, id + 1, ((other + 1) << 2) | (k + 1));
    }
    ore_reset(o, i, k);
    eng_set_state(o, piece, c->s_wait);
  }
  return 1;   /* the beam does not pass a hit ore */
}

static int cm_on_hit(Oracle* o, int target, int hitter, int hit) {
  Coop* c = cm(o);
  const Piece* t = &o->pieces[target];
  int blocked = 0;
  if (c->state_hit_block[t->state] & (1u << hit)) blocked = 1; /* BeamBlocker */
  if (t->kind == MPK_KIND_ORE && hit == c->hit_mine) {
    /* GameObject:_onHit: every component's onHit runs, any `true` blocks (game_object.lua:287-296) */
    for (int k = 0; k < CM_TYPES; ++k)
      if (ore_on_hit(o, t->index, k, o->pieces[hitter].index)) blocked = 1;
  }
  return blocked;
}

static void cm_on_enter(Oracle* o, int target, int entering, int contact) {
  (void)o; (void)target; (void)entering; (void)contact;   /* nothing reacts to a contact */
}

static void cm_on_state_change(Oracle* o, int piece, int old_state) {
  const Piece* p = &o->pieces[piece];
  if (p->kind == MPK_KIND_AVATAR) {
    int pl = p->index; /* Avatar:onStateChange (avatar_library.lua:430-453) */
    if (old_state == o->wait_state[pl] && p->state == o->alive_state[pl]) {
      o->freeze_counter[pl] = 0;
      o->removal_counter[pl] = 0;
    }
  }
}

const SubstrateVtbl kCoopVtbl = {
    cm_on_enter, cm_on_hit, cm_on_state_change,
    cm_sim_update, cm_run_updaters, cm_start,
};
