/* ORACLE — test infrastructure only.  See engine.h.
 *
 * collaborative_cooking rules: restatement of the reference's Lua components
 *   lua/levels/collaborative_cooking/components.lua  (InteractBeam :29-113, Container :116-181,
 *                                                     Inventory :184-277, Receiver :280-333,
 *                                                     CookingPot :336-474,
 *                                                     LoadingBarVisualiser :477-517)
 *   lua/modules/avatar_library.lua:155-203  (Avatar move: connected objects turn with it)
 * with kwargs from configs/substrates/collaborative_cooking.py (in the pack).
 *
 * An inventory is a piece whose STATE is the item it holds; getHeldItem reads the piece's
 * state as the engine has it — a setState queued by an earlier hit of the same flush is not
 * seen — while a pot's content, its `cooked` flag and a container's `usedThisStep` are Lua
 * variables and change at once.
 */
#include <stdlib.h>
#include <string.h>

#include "../include/mp_pack.h"
#include "engine.h"

enum { ACT_MOVE = 0, ACT_TURNA = 1, ACT_INTERACT = 2 };
enum { ITEM_EMPTY = 0, ITEM_TOMATO = 1, ITEM_DISH = 2, ITEM_SOUP = 3 };

typedef struct {
  int n_cont, n_recv, n_pot;
  const int32_t *cont_cells, *cont_i32, *recv_cells, *recv_i32, *pot_cells, *pot_states, *bar_states,
      *hits;
  const double* recv_f64;
  int s_wait, s_plain0, s_off0, s_dir0;
  int cooldown, cooking_time, bar_interval;
  double pot_reward;
  int *cont_inv, *pot_piece, *pot_bar;     /* pieces: inventory over container i; pot k; its bar */
  int av_inv[ORC_MAX_PLAYERS];             /* the inventory connected to avatar p */
  uint8_t* used;                           /* Container._usedThisStep */
  int *pot_count, *pot_time, *pot_cooked;  /* CookingPot._containedItems (a count), _currentCookingTime, _cooked */
  int pot_first;                           /* a pot is created before the first container (updater order) */
} Cook;

static Cook* ck(const Oracle* o) { return (Cook*)o->sub_state; }

void* cook_create(Oracle* o) {
  Cook* c = (Cook*)calloc(1, sizeof(Cook));
  uint64_t n;
  c->cont_cells = (const int32_t*)mpk_find(o->pack, "cc_container_cells", &n, 0); c->n_cont = (int)n;
  c->cont_i32 = (const int32_t*)mpk_find(o->pack, "cc_container_i32", &n, 0);
  c->recv_cells = (const int32_t*)mpk_find(o->pack, "cc_receiver_cells", &n, 0); c->n_recv = (int)n;
  c->recv_i32 = (const int32_t*)mpk_find(o->pack, "cc_receiver_i32", &n, 0);
  c->recv_f64 = (const double*)mpk_find(o->pack, "cc_receiver_f64", &n, 0);
  c->pot_cells = (const int32_t*)mpk_find(o->pack, "cc_pot_cells", &n, 0); c->n_pot = (int)n;
  c->pot_states = (const int32_t*)mpk_find(o->pack, "cc_pot_states", &n, 0);
  c->bar_states = (const int32_t*)mpk_find(o->pack, "cc_bar_states", &n, 0);
  c->hits = (const int32_t*)mpk_find(o->pack, "cc_hits", &n, 0);
  const int32_t* st = (const int32_t*)mpk_find(o->pack, "cc_inv_states", &n, 0);
  c->s_wait = st[0]; c->s_plain0 = st[1]; c->s_off0 = st[2]; c->s_dir0 = st[3];
  const int32_t* ci = (const int32_t*)mpk_find(o->pack, "cc_i32", &n, 0);
  c->cooldown = ci[0]; c->cooking_time = ci[1]; c->bar_interval = ci[2];
  c->pot_reward = ((const double*)mpk_find(o->pack, "cc_f64", &n, 0))[0];
  c->cont_inv = (int*)calloc((size_t)c->n_cont + 1, sizeof(int));
  c->used = (uint8_t*)calloc((size_t)c->n_cont + 1, 1);
  c->pot_piece = (int*)calloc((size_t)c->n_pot + 1, sizeof(int));
  c->pot_bar = (int*)calloc((size_t)c->n_pot + 1, sizeof(int));
  c->pot_count = (int*)calloc((size_t)c->n_pot + 1, sizeof(int));
  c->pot_time = (int*)calloc((size_t)c->n_pot + 1, sizeof(int));
  c->pot_cooked = (int*)calloc((size_t)c->n_pot + 1, sizeof(int));
  c->pot_first = 0;
  /* A19.  Every avatar's inventory starts at the Transform default (0, 0) without a layer
   * (collaborative_cooking.py:412-440,880-896); Inventory:_placeAtCorrectLocation gives it its
   * 'empty' state THERE before it teleports it onto the avatar (components.lua:209-221) — and in
   * three of the seven layouts (0, 0) is a counter whose own inventory already holds that cell of
   * the layer.  The substrates work in the reference, so the state change cannot be what fails. */
  o->opt_set_state_lifts = 1;
  for (int i = 0; i < o->nobj; ++i) {   /* which registers its 140 updater first */
    const int kind = o->objects[4 * i];
    if (kind == MPK_KIND_POT) { c->pot_first = 1; break; }
    if (kind == MPK_KIND_CONTAINER) break;
  }
  return c;
}

void cook_destroy(void* s) {
  Cook* c = (Cook*)s;
  if (!c) return;
  free(c->cont_inv); free(c->used); free(c->pot_piece); free(c->pot_bar);
  free(c->pot_count); free(c->pot_time); free(c->pot_cooked); free(c);
}

int cook_cooldown(const Oracle* o) { return ck(o)->cooldown > 0 ? ck(o)->cooldown : 1; }

/* Inventory:getHeldItem (components.lua:245-252): the state's name without '_offset' */
static int held_item(const Oracle* o, int piece) {
  const Cook* c = ck(o);
  const int s = o->pieces[piece].state;
  if (s >= c->s_plain0 && s < c->s_plain0 + 4) return s - c->s_plain0;
  if (s >= c->s_off0 && s < c->s_off0 + 4) return s - c->s_off0;
  return -1;   /* 'wait' */
}
/* Inventory:setHeldItem (:254-260) */
static void set_held(Oracle* o, int piece, int item, int of_player) {
  eng_set_state(o, piece, (of_player ? ck(o)->s_off0 : ck(o)->s_plain0) + item);
}

/* what the state dump carries beyond the engine's pieces: an avatar's inventory shows its
 * facing in the state id (the pack's pseudo-states: the GPU engine keeps no orientation of
 * non-avatar pieces); the pots' cooking times */
void cook_dump(const Oracle* o, uint8_t* grid, int32_t* glob) {
  const Cook* c = ck(o);
  for (int p = 0; p < o->P; ++p) {
    const Piece* pc = &o->pieces[c->av_inv[p]];
    const int layer = o->state_layer[pc->state];
    if (layer < 0) continue;
    const int item = pc->state - c->s_off0;
    if (item < 0 || item >= 4 || pc->orient == ORIENT_N) continue;
    grid[((size_t)layer * o->H + pc->y) * o->W + pc->x] = (uint8_t)(c->s_dir0 + (pc->orient - 1) * 4 + item);
  }
  uint32_t times = 0;
  for (int k = 0; k < c->n_pot; ++k) times += (uint32_t)c->pot_time[k] * (uint32_t)(k + 1);
  glob[3] = 0; glob[5] = (int32_t)times;
}

static void add_reward(Oracle* o, int p, double amount) {
  if (o->pieces[o->avatar_piece[p]].state != o->wait_state[p]) o->reward[p] += amount;
}

static int piece_at_cell(const Oracle* o, int kind, int cell) {
  for (int i = 0; i < o->npieces; ++i)
    if (o->pieces[i].kind == kind && o->pieces[i].y * o->W + o->pieces[i].x == cell) return i;
  return -1;
}

static void ck_start(Oracle* o) {
  Cook* c = ck(o);
  /* the container inventories sit where their containers are (kept off the grid until now:
   * A18), the avatars' are the LAST P inventory pieces, in avatar order */
  int n_inv = 0, inv[1024];
  for (int i = 0; i < o->npieces && n_inv < 1024; ++i)
    if (o->pieces[i].kind == MPK_KIND_INVENTORY) inv[n_inv++] = i;
  for (int i = 0; i < c->n_cont; ++i) {
    c->cont_inv[i] = -1;
    for (int j = 0; j < n_inv - o->P_pack; ++j)
      if (o->pieces[inv[j]].y * o->W + o->pieces[inv[j]].x == c->cont_cells[i]) c->cont_inv[i] = inv[j];
    c->used[i] = 0;
    /* Inventory:postStart: setState(emptyState); Container:attachInventory: setHeldItem(startingItem) */
    eng_set_state(o, c->cont_inv[i], c->s_plain0 + ITEM_EMPTY);
    set_held(o, c->cont_inv[i], c->cont_i32[2 * i], 0);
  }
  for (int k = 0; k < c->n_pot; ++k) {
    c->pot_piece[k] = piece_at_cell(o, MPK_KIND_POT, c->pot_cells[k]);
    c->pot_bar[k] = piece_at_cell(o, MPK_KIND_LOADING_BAR, c->pot_cells[k]);
    c->pot_count[k] = 0; c->pot_time[k] = 0; c->pot_cooked[k] = 0;   /* CookingPot:reset */
  }
  for (int p = 0; p < o->P; ++p) {
    /* Inventory:_placeAtCorrectLocation (:209-221): a state with a layer first, then onto the
     * avatar, connect, face the way it faces */
    const int piece = inv[n_inv - o->P_pack + p];
    const Piece* av = &o->pieces[o->avatar_piece[p]];
    c->av_inv[p] = piece;
    eng_set_state(o, piece, c->s_plain0 + ITEM_EMPTY);
    eng_teleport(o, piece, av->x, av->y);
    eng_connect(o, o->avatar_piece[p], piece);
    eng_set_orientation(o, piece, av->orient);
  }
  /* InteractBeam:reset: _coolingTimer = 0 (zap_timer, cleared by the episode start) */
}

static void ck_sim_update(Oracle* o) {
  for (int p = 0; p < o->P; ++p) o->reward[p] = 0.0; /* Avatar:preUpdate */
  for (int p = 0; p < o->P; ++p) {
    /* Avatar:update (avatar_library.lua:334-355); AvatarCumulants:update is debug only */
    if (o->freeze_counter[p] == 1) o->movement_allowed[p] = 1;
    if (o->freeze_counter[p] > 0) o->freeze_counter[p]--;
    if (o->removal_counter[p] == 1) eng_set_state(o, o->avatar_piece[p], o->wait_state[p]);
    if (o->removal_counter[p] > 0) o->removal_counter[p]--;
  }
}

static void tick_containers(Oracle* o) {
  Cook* c = ck(o);
  eng_trace(o, 140, "Container.tick");
  for (int i = 0; i < c->n_cont; ++i) c->used[i] = 0;
}
static void tick_pots(Oracle* o) {
  Cook* c = ck(o);
  /* CookingPot tickPotFn (components.lua:452-470) */
  eng_trace(o, 140, "CookingPot.tickPotFn");
  for (int k = 0; k < c->n_pot; ++k)
    if (c->pot_count[k] == 3 && !c->pot_cooked[k]) {
      if (c->pot_time[k] == c->cooking_time) {
        c->pot_cooked[k] = 1;
        eng_set_state(o, c->pot_piece[k], c->pot_states[4]);
      }
      c->pot_time[k]++;
    }
}

static void ck_run_updaters(Oracle* o) {
  Cook* c = ck(o);
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
      if (o->pieces[c->av_inv[p]].leader == o->avatar_piece[p]) eng_turn(o, c->av_inv[p], turn);
    }
    if (move != 0) eng_move_rel(o, o->avatar_piece[p], move - 1);
  }
  /* 140, in registration order (A11): the explicit objects come first — the loading bars
   * (collaborative_cooking.py:727-754), then the avatars — and the map's objects after them */
  /* LoadingBarVisualiser tickLoadingBarFn (components.lua:495-512): the pot's time as the
   * PREVIOUS frame's tick left it */
  eng_trace(o, 140, "LoadingBarVisualiser.tickLoadingBarFn");
  for (int k = 0; k < c->n_pot; ++k) {
    const int idx = c->pot_time[k] / c->bar_interval;   /* floor(time / interval) + 1, 1-based */
    eng_set_state(o, c->pot_bar[k], c->bar_states[idx < 10 ? idx : 10]);
  }
  /* InteractBeam interact (components.lua:79-100) */
  eng_trace(o, 140, "InteractBeam.interact");
  for (int p = 0; p < P; ++p) order[p] = p;
  eng_shuffle(o, RS_SHUFFLE_ZAP, order, P);
  for (int i = 0; i < P; ++i) {
    int p = order[i];
    if (c->cooldown < 0) continue;
    if (o->zap_timer[p] > 0) o->zap_timer[p]--;
    else if (o->action[p][ACT_INTERACT] == 1) {
      o->zap_timer[p] = c->cooldown;
      eng_hit_beam(o, o->avatar_piece[p], c->hits[p], 1, 0);
    }
  }
  if (c->pot_first) { tick_pots(o); tick_containers(o); }
  else { tick_containers(o); tick_pots(o); }
}

static int index_of(const int32_t* cells, int n, int cell) {
  for (int i = 0; i < n; ++i) if (cells[i] == cell) return i;
  return -1;
}

static int ck_on_hit(Oracle* o, int target, int hitter, int hit) {
  Cook* c = ck(o);
  (void)hit;   /* "Assume nothing will send a hit that doesn't also have InteractBeam" */
  const Piece* t = &o->pieces[target];
  if (o->pieces[hitter].kind != MPK_KIND_AVATAR) return 0;
  const int pl = o->pieces[hitter].index;
  const int inv = c->av_inv[pl];
  const int cell = t->y * o->W + t->x;
  if (t->kind == MPK_KIND_CONTAINER) {
    /* Container:onHit (components.lua:137-163) */
    const int i = index_of(c->cont_cells, c->n_cont, cell);
    if (i < 0 || c->used[i]) return 0;
    c->used[i] = 1;
    const int mine = held_item(o, inv), its = held_item(o, c->cont_inv[i]);
    if (its != ITEM_EMPTY && mine == ITEM_EMPTY) {
      set_held(o, inv, its, 1);
      if (!c->cont_i32[2 * i + 1]) set_held(o, c->cont_inv[i], ITEM_EMPTY, 0);
    } else if (its == ITEM_EMPTY && mine != ITEM_EMPTY) {
      set_held(o, inv, ITEM_EMPTY, 1);
      set_held(o, c->cont_inv[i], mine, 0);
    }
  } else if (t->kind == MPK_KIND_RECEIVER) {
    /* Receiver:onHit (components.lua:301-333) */
    const int j = index_of(c->recv_cells, c->n_recv, cell);
    const int mine = held_item(o, inv);
    if (j >= 0 && mine == c->recv_i32[2 * j]) {
      if (c->recv_i32[2 * j + 1]) {   /* every avatar of the 'players' group */
        for (int q = 0; q < o->P; ++q)
          if (o->pieces[o->avatar_piece[q]].state == o->alive_state[q]) add_reward(o, q, c->recv_f64[j]);
      } else {
        add_reward(o, pl, c->recv_f64[j]);
      }
      set_held(o, inv, ITEM_EMPTY, 1);
      eng_event(o, 17 /* receiver_accepted_item */, pl + 1, mine);
    }
  } else if (t->kind == MPK_KIND_POT) {
    /* CookingPot:onHit (components.lua:378-448) */
    const int k = index_of(c->pot_cells, c->n_pot, cell);
    if (k < 0) return 0;
    const int mine = held_item(o, inv);
    if (mine == ITEM_TOMATO && c->pot_count[k] < 3) {
      c->pot_count[k]++;
      add_reward(o, pl, c->pot_reward);
      set_held(o, inv, ITEM_EMPTY, 1);
      eng_event(o, 18 /* item_dropped_into_pot */, pl + 1, mine);
    } else if (mine == ITEM_DISH && c->pot_cooked[k]) {
      add_reward(o, pl, c->pot_reward);
      set_held(o, inv, ITEM_SOUP, 1);
      c->pot_count[k] = 0; c->pot_cooked[k] = 0; c->pot_time[k] = 0;
      eng_event(o, 19 /* cooked_food_collected_from_pot */, pl + 1, ITEM_SOUP);
    }
    if (!c->pot_cooked[k]) eng_set_state(o, c->pot_piece[k], c->pot_states[c->pot_count[k]]);
  }
  return 0;   /* (no onHit of this level returns anything: the beam is one cell long anyway) */
}

static void ck_on_enter(Oracle* o, int target, int entering, int contact) {
  (void)o; (void)target; (void)entering; (void)contact;   /* nothing reacts to a contact */
}

static void ck_on_state_change(Oracle* o, int piece, int old_state) {
  const Piece* p = &o->pieces[piece];
  if (p->kind == MPK_KIND_AVATAR) {
    int pl = p->index; /* Avatar:onStateChange (avatar_library.lua:430-453) */
    if (old_state == o->wait_state[pl] && p->state == o->alive_state[pl]) {
      o->freeze_counter[pl] = 0;
      o->removal_counter[pl] = 0;
    }
  }
}

const SubstrateVtbl kCookVtbl = {
    ck_on_enter, ck_on_hit, ck_on_state_change,
    ck_sim_update, ck_run_updaters, ck_start,
};
