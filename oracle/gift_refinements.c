/* ORACLE — test infrastructure only.  See engine.h.
 *
 * gift_refinements rules: restatement of the reference's Lua components
 *   lua/levels/gift_refinements/components.lua  (FixedRateRegrow :29-55, Pickable :57-90,
 *                                                GiftBeam :92-237, Inventory :239-353,
 *                                                TokenTracker :355-393)
 *   lua/modules/component_library.lua:907-948  (StochasticIntervalEpisodeEnding),
 *                                    :667-685  (BeamBlocker)
 *   lua/modules/avatar_library.lua             (Avatar; no Zapper in this level)
 * with kwargs from configs/substrates/gift_refinements.py (in the pack).
 *
 * Unlike coop_mining's, this level's FixedRateRegrow is a component update(): it runs in
 * BaseSimulation:update (token objects in creation order, after the avatars), draws only
 * for a token in its wait state, and its setState is queued AHEAD of every updater's events.
 * Inventory:addTokens RETURNS THE NEW COUNT, not the amount added (:307-318): that count is
 * what GiftBeam:onHit files as the gift's "actual" amount and reports in the event.
 */
#include <stdlib.h>
#include <string.h>

#include "../include/mp_pack.h"
#include "engine.h"

enum { ACT_MOVE = 0, ACT_TURNA = 1, ACT_GIFT = 2, ACT_CONSUME = 3 };
enum { GR_MAX_TYPES = 3 };

typedef struct {
  int n_token;
  int* token_piece;               /* token pieces in creation order */
  int s_wait, s_live;
  int cooldown, beam_length, beam_radius, hit_gift;
  int capacity, ntypes, multiplier, consume_cooldown;
  int inventory[ORC_MAX_PLAYERS][GR_MAX_TYPES];   /* Inventory.inventory */
  int consume_timer[ORC_MAX_PLAYERS];             /* Inventory._consumeCooldownTimer (<= 0: ready) */
  const double* reward;           /* [P][2]: per hit, per refined gift; then picking */
  double pick_reward;
  const uint64_t* thr;            /* regrow, episode end */
  const uint32_t* state_hit_block;
  int ee_min_frames, ee_interval, ee_t;
} Gift;

static Gift* gr(const Oracle* o) { return (Gift*)o->sub_state; }

void* gift_create(Oracle* o) {
  Gift* c = (Gift*)calloc(1, sizeof(Gift));
  uint64_t n;
  const int32_t* st = (const int32_t*)mpk_find(o->pack, "gr_states", &n, 0);
  const int32_t* ci = (const int32_t*)mpk_find(o->pack, "gr_i32", &n, 0);
  c->s_wait = st[0]; c->s_live = st[1];
  c->cooldown = ci[0]; c->beam_length = ci[1]; c->beam_radius = ci[2]; c->hit_gift = ci[3];
  c->ee_min_frames = ci[4]; c->ee_interval = ci[5];
  c->capacity = ci[6]; c->ntypes = ci[7]; c->multiplier = ci[8]; c->consume_cooldown = ci[9];
  c->reward = (const double*)mpk_find(o->pack, "gr_f64", &n, 0);
  c->pick_reward = c->reward[2 * o->P_pack];
  c->thr = (const uint64_t*)mpk_find(o->pack, "gr_thr", &n, 0);
  c->state_hit_block = (const uint32_t*)mpk_find(o->pack, "state_hit_block", &n, 0);
  mpk_find(o->pack, "token_cells", &n, 0);
  c->n_token = (int)n;
  c->token_piece = (int*)calloc((size_t)c->n_token, sizeof(int));
  return c;
}

void gift_destroy(void* s) {
  Gift* c = (Gift*)s;
  if (!c) return;
  free(c->token_piece); free(c);
}

/* what the state dump carries of the Lua-side variables: live tokens; per avatar the
 * inventory (4 bits a type) and whether the consumption timer is running */
void gift_dump(const Oracle* o, int32_t* avat, int32_t* glob) {
  const Gift* c = gr(o);
  int live = 0;
  for (int i = 0; i < c->n_token; ++i) live += o->pieces[c->token_piece[i]].state == c->s_live;
  glob[3] = live;
  for (int p = 0; p < o->P; ++p) {
    avat[8 * p + 5] = c->consume_timer[p] > 0 ? c->consume_timer[p] : 0;
    avat[8 * p + 7] = c->inventory[p][0] | (c->inventory[p][1] << 4) | (c->inventory[p][2] << 8);
  }
}

int gift_cooldown(const Oracle* o) { return gr(o)->cooldown; }
int gift_num_types(const Oracle* o) { return gr(o)->ntypes; }
void gift_inventory(const Oracle* o, int p, double* out) {
  for (int k = 0; k < gr(o)->ntypes; ++k) out[k] = (double)gr(o)->inventory[p][k];
}

static void add_reward(Oracle* o, int p, double amount) {
  /* Avatar:addReward with skipWaitStateRewards (avatar_library.lua:362-376) */
  if (o->pieces[o->avatar_piece[p]].state != o->wait_state[p]) o->reward[p] += amount;
}

/* Inventory:addTokens (components.lua:307-318): returns the NEW count */
static int add_tokens(Gift* c, int p, int type, int amount) {
  int v = c->inventory[p][type] + amount;
  if (v > c->capacity) v = c->capacity;
  c->inventory[p][type] = v;
  return v;
}

static void gr_start(Oracle* o) {
  Gift* c = gr(o);
  int n = 0;
  for (int i = 0; i < o->npieces; ++i)
    if (o->pieces[i].kind == MPK_KIND_TOKEN) c->token_piece[n++] = i;
  c->ee_t = 1;
  /* Inventory:reset / :start, GiftBeam:start (zap_timer is cleared by the episode start) */
  memset(c->inventory, 0, sizeof(c->inventory));
  memset(c->consume_timer, 0, sizeof(c->consume_timer));
}

/* BaseSimulation:update (base_simulation.lua:476-486): preUpdate on all, then update on
 * all, objects in creation order: scene, avatars, map objects. */
static void gr_sim_update(Oracle* o) {
  Gift* c = gr(o);
  for (int p = 0; p < o->P; ++p) o->reward[p] = 0.0; /* Avatar:preUpdate; TokenTracker:preUpdate */
  c->ee_t++; /* StochasticIntervalEpisodeEnding:update */
  for (int p = 0; p < o->P; ++p) {
    /* Avatar:update (avatar_library.lua:334-355) */
    if (o->freeze_counter[p] == 1) o->movement_allowed[p] = 1;
    if (o->freeze_counter[p] > 0) o->freeze_counter[p]--;
    if (o->removal_counter[p] == 1) eng_set_state(o, o->avatar_piece[p], o->wait_state[p]);
    if (o->removal_counter[p] > 0) o->removal_counter[p]--;
    /* Inventory:update (components.lua:328-350) */
    if (o->action[p][ACT_CONSUME] == 1 && c->consume_timer[p] <= 0) {
      int amount = 0;
      for (int k = 0; k < c->ntypes; ++k) { amount += c->inventory[p][k]; c->inventory[p][k] = 0; }
      add_reward(o, p, (double)amount);
      c->consume_timer[p] = c->consume_cooldown;
    }
    c->consume_timer[p]--;
    if (c->consume_timer[p] < 0) c->consume_timer[p] = 0;   /* (Lua counts down for ever; <= 0 is "ready") */
    /* GiftBeam:update (components.lua:222-226) */
    if (o->zap_timer[p] > 0) o->zap_timer[p]--;
  }
  /* FixedRateRegrow:update, token by token (components.lua:45-55): one draw per WAITING
   * token; the avatar is looked for where it stands now, before this frame's moves */
  const int upper = o->avatar_layer;
  for (int i = 0; i < c->n_token; ++i) {
    const Piece* pc = &o->pieces[c->token_piece[i]];
    if (pc->state != c->s_wait) continue;
    if (eng_u53(o, eng_draw(o, RS_REGROW, (uint32_t)i)) >= c->thr[0]) continue;
    if (eng_cell(o, upper, pc->x, pc->y) >= 0) continue;   /* queryPosition('upperPhysical') */
    eng_set_state(o, c->token_piece[i], c->s_live);
  }
}

static void gr_run_updaters(Oracle* o) {
  Gift* c = gr(o);
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
  /* 140: GiftBeam gift (components.lua:186-211) */
  eng_trace(o, 140, "GiftBeam.gift");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_ZAP, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i];
    if (o->pieces[o->avatar_piece[p]].state != o->alive_state[p]) continue;
    if (o->action[p][ACT_GIFT] == 1 && o->zap_timer[p] <= 0) {
      o->zap_timer[p] = c->cooldown;
      eng_hit_beam(o, o->avatar_piece[p], c->hit_gift, c->beam_length, c->beam_radius);
    }
  }
  /* 100: StochasticIntervalEpisodeEnding (component_library.lua:927-940) */
  eng_trace(o, 100, "StochasticIntervalEpisodeEnding.maybeEndEpisode");
  if (eng_frames(o, 0) >= c->ee_min_frames && c->ee_t % c->ee_interval == 0) {
    if (eng_u53(o, eng_draw(o, RS_EPISODE_END, 0)) < c->thr[1]) o->continue_flag = 0;
  }
}

/* GiftBeam:onHit (components.lua:135-184) */
static int gift_on_hit(Oracle* o, int hit_player, int hitter_player) {
  Gift* c = gr(o);
  const double amount = c->reward[2 * hitter_player];
  add_reward(o, hitter_player, amount);
  /* Inventory:getHighestTypeAvailable */
  int src = -1;
  for (int k = 0; k < c->ntypes; ++k) if (c->inventory[hitter_player][k] > 0) src = k;
  if (src >= 0) {
    int dst_amount = c->multiplier, dst = src + 1;
    if (dst >= c->ntypes) { dst = c->ntypes - 1; dst_amount = 1; }   /* the most refined: passed on as it is */
    else add_reward(o, hitter_player, c->reward[2 * hitter_player + 1]);   /* amount * successfulGiftReward */
    /* Inventory:removeTokens(srcType, 1) */
    c->inventory[hitter_player][src] -= 1;
    const int actual = add_tokens(c, hit_player, dst, dst_amount);
    /* events:add("gift", ...): gifter | source type; recipient | the count it now holds */
    eng_event(o, 16, (hitter_player + 1) | ((src + 1) << 4), (hit_player + 1) | (actual << 4));
  }
  return 1;   /* the beam does not pass a hit player */
}

static int gr_on_hit(Oracle* o, int target, int hitter, int hit) {
  Gift* c = gr(o);
  const Piece* t = &o->pieces[target];
  int blocked = 0;
  if (c->state_hit_block[t->state] & (1u << hit)) blocked = 1; /* BeamBlocker */
  if (t->kind == MPK_KIND_AVATAR && hit == c->hit_gift)
    if (gift_on_hit(o, t->index, o->pieces[hitter].index)) blocked = 1;
  return blocked;
}

/* Pickable:onEnter (components.lua:74-90) */
static void gr_on_enter(Oracle* o, int target, int entering, int contact) {
  Gift* c = gr(o);
  (void)contact; /* the only contact in this level is 'avatar' */
  const Piece* t = &o->pieces[target];
  const Piece* e = &o->pieces[entering];
  if (t->kind != MPK_KIND_TOKEN || e->kind != MPK_KIND_AVATAR) return;
  if (t->state != c->s_live) return;
  add_reward(o, e->index, c->pick_reward);
  add_tokens(c, e->index, 0, 1);
  eng_set_state(o, target, c->s_wait);
}

static void gr_on_state_change(Oracle* o, int piece, int old_state) {
  const Piece* p = &o->pieces[piece];
  if (p->kind == MPK_KIND_AVATAR) {
    int pl = p->index; /* Avatar:onStateChange (avatar_library.lua:430-453) */
    if (old_state == o->wait_state[pl] && p->state == o->alive_state[pl]) {
      o->freeze_counter[pl] = 0;
      o->removal_counter[pl] = 0;
    }
  }
}

const SubstrateVtbl kGiftVtbl = {
    gr_on_enter, gr_on_hit, gr_on_state_change,
    gr_sim_update, gr_run_updaters, gr_start,
};
