/* ORACLE — test infrastructure only.  See engine.h.
 *
 * coins rules: restatement of the reference's Lua components
 *   lua/levels/coins/components.lua   (PlayerCoinType, Coin, ChoiceCoinRegrow,
 *     GlobalCoinCollectionTracker, Role, PartnerTracker)
 *   lua/modules/component_library.lua:907-948 (StochasticIntervalEpisodeEnding)
 *   lua/modules/avatar_library.lua            (Avatar; there is no Zapper)
 * with kwargs from configs/substrates/coins.py (in the pack: every map its
 * generator can draw, one per world, and one instance of the coin colours).
 */
#include <stdlib.h>
#include <string.h>

#include "../include/mp_pack.h"
#include "engine.h"

enum { ACT_MOVE = 0, ACT_TURNA = 1 };

typedef struct {
  int n_coin;
  int* coin_piece;                 /* piece of coin site i of the pack (its per-kind index), or -1:
                                      the site is not part of this world's map (mapAlternatives) */
  int s_coin[2], s_wait;
  int player_type[ORC_MAX_PLAYERS];        /* PlayerCoinType: index into s_coin */
  double rew[ORC_MAX_PLAYERS][4];  /* self match, self mismatch, other match, other mismatch */
  uint64_t thr_regrow, thr_ee;
  int ee_min_frames, ee_interval, ee_t;
  /* PartnerTracker.partnerCollectedMismatch (components.lua:281-328) */
  int partner_mismatch[ORC_MAX_PLAYERS];
  int32_t alive[ORC_MAX_PLAYERS];  /* the avatars' alive states in this world's colours */
} Coins;

static Coins* co(const Oracle* o) { return (Coins*)o->sub_state; }

void* coins_create(Oracle* o) {
  Coins* c = (Coins*)calloc(1, sizeof(Coins));
  uint64_t n;
  const int32_t* st = (const int32_t*)mpk_find(o->pack, "co_states", &n, 0);
  const int32_t* ci = (const int32_t*)mpk_find(o->pack, "co_i32", &n, 0);
  const double* cf = (const double*)mpk_find(o->pack, "co_f64", &n, 0);
  const uint64_t* thr = (const uint64_t*)mpk_find(o->pack, "co_thr", &n, 0);
  c->s_coin[0] = st[0]; c->s_coin[1] = st[1]; c->s_wait = st[2];
  for (int p = 0; p < o->P; ++p) {
    c->player_type[p] = ci[p];
    for (int k = 0; k < 4; ++k) c->rew[p][k] = cf[4 * p + k];
  }
  c->ee_min_frames = ci[o->P]; c->ee_interval = ci[o->P + 1];
  c->thr_regrow = thr[0]; c->thr_ee = thr[1];
  /* The two coin colours of this world: coins.py:500 draws random.sample(COIN_PALETTES,
   * k=2) when the environment is built — player 1 and coin type A wear the first,
   * player 2 and type B the second.  One per-world draw of the 20 ordered pairs
   * (the map choices' stream, index 0x10000, no episode in the counter); the pack
   * holds the coin's state and each avatar's alive state per colour. */
  const int32_t* cc = (const int32_t*)mpk_find(o->pack, "co_colour_coin", &n, 0);
  const int32_t* ca = (const int32_t*)mpk_find(o->pack, "co_colour_alive", &n, 0);
  if (cc && ca) {
    int pair = (int)eng_bounded(o, 
        philox4x32_10(0x10000u, RS_MAP_CHOICE, 0u, 0xffffffffu, (uint32_t)o->world_seed,
                      (uint32_t)(o->world_seed >> 32)), 20u);
    int a = pair >> 2, r = pair & 3, b = r + (r >= a ? 1 : 0);
    c->s_coin[0] = cc[a]; c->s_coin[1] = cc[b];
    c->alive[0] = ca[a]; c->alive[1] = ca[5 + b];
    o->alive_state = c->alive;
  }
  mpk_find(o->pack, "coin_cells", &n, 0);
  c->n_coin = (int)n;
  c->coin_piece = (int*)calloc((size_t)c->n_coin, sizeof(int));
  return c;
}

void coins_destroy(void* s) {
  Coins* c = (Coins*)s;
  if (!c) return;
  free(c->coin_piece); free(c);
}

int coins_live(const Oracle* o) {
  const Coins* c = co(o);
  int n = 0;
  for (int i = 0; i < c->n_coin; ++i)
    n += c->coin_piece[i] >= 0 && o->pieces[c->coin_piece[i]].state != c->s_wait;
  return n;
}

/* "N.MISMATCHED_COIN_COLLECTED_BY_PARTNER" (configs/substrates/coins.py:352-359) */
double coins_partner_mismatch(const Oracle* o, int p) { return (double)co(o)->partner_mismatch[p]; }

static void co_start(Oracle* o) {
  Coins* c = co(o);
  for (int i = 0; i < c->n_coin; ++i) c->coin_piece[i] = -1;
  for (int i = 0; i < o->npieces; ++i)
    if (o->pieces[i].kind == MPK_KIND_COIN) c->coin_piece[o->pieces[i].index] = i;
  c->ee_t = 1;
  for (int p = 0; p < o->P; ++p) c->partner_mismatch[p] = 0; /* PartnerTracker:reset */
}

/* BaseSimulation:update (base_simulation.lua:476-486): preUpdate on all, then
 * update on all. */
static void co_sim_update(Oracle* o) {
  Coins* c = co(o);
  for (int p = 0; p < o->P; ++p) {
    o->reward[p] = 0.0;          /* Avatar:preUpdate */
    c->partner_mismatch[p] = 0;  /* PartnerTracker:preUpdate (components.lua:300-303) */
  }
  c->ee_t++; /* StochasticIntervalEpisodeEnding:update */
  for (int p = 0; p < o->P; ++p) { /* Avatar:update (avatar_library.lua:334-355) */
    if (o->freeze_counter[p] == 1) o->movement_allowed[p] = 1;
    if (o->freeze_counter[p] > 0) o->freeze_counter[p]--;
    if (o->removal_counter[p] == 1) eng_set_state(o, o->avatar_piece[p], o->wait_state[p]);
    if (o->removal_counter[p] > 0) o->removal_counter[p]--;
  }
}

static void co_run_updaters(Oracle* o) {
  Coins* c = co(o);
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
  /* 100: StochasticIntervalEpisodeEnding (component_library.lua:927-940) */
  eng_trace(o, 100, "StochasticIntervalEpisodeEnding.maybeEndEpisode");
  if (eng_frames(o, 0) >= c->ee_min_frames && c->ee_t % c->ee_interval == 0) {
    if (eng_u53(o, eng_draw(o, RS_EPISODE_END, 0)) < c->thr_ee) o->continue_flag = 0;
  }
  /* 100: ChoiceCoinRegrow (components.lua:190-201): state = waitState,
   * probability = regrowRate (A12: one draw per waiting piece), then
   * random:choice(liveStates): a second draw of the same piece. */
  eng_trace(o, 100, "ChoiceCoinRegrow.regrow");
  for (int i = 0; i < c->n_coin; ++i) {
    int piece = c->coin_piece[i];
    if (piece < 0 || o->pieces[piece].state != c->s_wait) continue;
    if (eng_u53(o, eng_draw(o, RS_REGROW, (uint32_t)i)) >= c->thr_regrow) continue;
    int k = (int)eng_bounded(o, eng_draw(o, RS_COIN_CHOICE, (uint32_t)i), 2u);
    eng_set_state(o, piece, c->s_coin[k]);
  }
}

static int co_on_hit(Oracle* o, int target, int hitter, int hit) {
  (void)o; (void)target; (void)hitter; (void)hit; /* no beams in this level */
  return 0;
}

static void co_on_enter(Oracle* o, int target, int entering, int contact) {
  Coins* c = co(o);
  (void)contact; /* the only contact is 'avatar' */
  const Piece* t = &o->pieces[target];
  if (t->kind != MPK_KIND_COIN || t->state == c->s_wait) return;
  /* Coin:onEnter (components.lua:93-170) */
  int p = o->pieces[entering].index;
  int coin_type = t->state == c->s_coin[1];
  int match = coin_type == c->player_type[p];
  o->reward[p] += c->rew[p][match ? 0 : 1];
  for (int q = 0; q < o->P; ++q)        /* Coin:rewardOthers */
    if (q != p) o->reward[q] += c->rew[p][match ? 2 : 3];
  /* PartnerTracker:reportMatch / reportMismatch set the PARTNER's tracker */
  if (!match) c->partner_mismatch[p == 0 ? 1 : 0] = 1;
  eng_event(o, 10 /* coin_consumed (components.lua:151-154) */, p + 1,
            (c->player_type[p] << 1) | coin_type);
  eng_set_state(o, target, c->s_wait);
}

static void co_on_state_change(Oracle* o, int piece, int old_state) {
  const Piece* p = &o->pieces[piece];
  if (p->kind == MPK_KIND_AVATAR) {
    int pl = p->index; /* Avatar:onStateChange (avatar_library.lua:430-453) */
    if (old_state == o->wait_state[pl] && p->state == o->alive_state[pl]) {
      o->freeze_counter[pl] = 0;
      o->removal_counter[pl] = 0;
    }
  }
}

const SubstrateVtbl kCoinsVtbl = {
    co_on_enter, co_on_hit, co_on_state_change,
    co_sim_update, co_run_updaters, co_start,
};
