/* ORACLE — test infrastructure only.  See engine.h for scope and provenance.
 *
 * Grid engine restatement (dmlab2d==1.0.0 `system.grid_world`, absent from the
 * reference tree).  Each function cites the reference call site / document it
 * follows; behaviours that nothing in the reference pins are marked A<n>
 * (DESIGN.md "engine unknowns").
 */
#include "engine.h"
#include "mt19937_64.h"

#include <stdlib.h>
#include <string.h>

#include "../include/mp_pack.h"

static const int kDx[4] = {0, 1, 0, -1}; /* N E S W: N = decreasing y */
static const int kDy[4] = {-1, 0, 1, 0}; /* (component_library.lua:379-386) */

static inline int cell_index(const Oracle* o, int layer, int x, int y) {
  return (layer * o->H + y) * o->W + x;
}

int eng_cell(const Oracle* o, int layer, int x, int y) {
  return o->cell[cell_index(o, layer, x, y)];
}

int eng_on_grid(const Oracle* o, int piece) {
  return o->state_layer[o->pieces[piece].state] >= 0;
}

/* grid:frames(piece): frames since the last state change
 * (component_library.lua:440-442, updater_registry.lua:53-56). */
int eng_frames(const Oracle* o, int piece) {
  return o->frame - o->pieces[piece].change_frame;
}

void eng_trace(Oracle* o, int priority, const char* tag) {
  if (o->trace_n < ORC_MAX_TRACE) {
    o->trace[o->trace_n].priority = priority;
    o->trace[o->trace_n].tag = tag;
    o->trace_n++;
  }
}

PhiloxOut eng_draw(const Oracle* o, int stream, uint32_t index) {
  /* A10: counter = {index, stream, step, episode}, key = world seed */
  if (o->opt_serial_rng) {   /* A10s: nothing is consumed until a call site takes a value */
    PhiloxOut none = {{0, 0, 0, 0}};
    return none;
  }
  return philox4x32_10(index, (uint32_t)stream, (uint32_t)o->step, o->ep, o->k0,
                       o->k1);
}

struct Mt64State { Mt64 g; };

/* random:seed(seed) of api:start (api_factory.lua:89).  Which seed dmlab2d's Python
 * layer hands to episode k is not in the reference tree: world seed + k here. */
void eng_reseed(Oracle* o) {
  if (!o->mt) o->mt = (struct Mt64State*)calloc(1, sizeof(struct Mt64State));
  mt64_seed(&o->mt->g, o->world_seed + (uint64_t)o->ep);
}
uint64_t eng_u53(const Oracle* o, PhiloxOut d) {
  return o->opt_serial_rng ? mt64_u53(&o->mt->g) : philox_u53(d);
}
uint32_t eng_bounded(const Oracle* o, PhiloxOut d, uint32_t n) {
  return o->opt_serial_rng ? (uint32_t)mt64_bounded(&o->mt->g, n, o->opt_serial_int_method)
                           : philox_bounded(d, n);
}
uint32_t eng_pick4(const Oracle* o, PhiloxOut d) {
  return o->opt_serial_rng ? (uint32_t)mt64_bounded(&o->mt->g, 4, o->opt_serial_int_method)
                           : (d.x[3] & 3u);
}

/* A1: the engine visits the pieces of an updater group in a freshly shuffled
 * order every frame.  Forward Fisher-Yates, one draw per position. */
void eng_shuffle(const Oracle* o, int stream, int* items, int n) {
  if (!o->opt_shuffle_order) return; /* A1 off: creation (player index) order */
  if (o->opt_serial_rng && o->opt_serial_shuffle_back) {   /* A10s: from the back */
    for (int i = n - 1; i > 0; --i) {
      int j = (int)eng_bounded(o, eng_draw(o, stream, (uint32_t)i), (uint32_t)(i + 1));
      int t = items[i]; items[i] = items[j]; items[j] = t;
    }
    return;
  }
  for (int i = 0; i + 1 < n; ++i) {
    int j = i + (int)eng_bounded(o, eng_draw(o, stream, (uint32_t)i),
                                 (uint32_t)(n - i));
    int t = items[i];
    items[i] = items[j];
    items[j] = t;
  }
}

/* All mutators are queued until the engine's event phase
 * (docs/advanced.md:24-31; game_object_test.lua:182-188). */
void eng_queue(Oracle* o, int kind, int piece, int a, int b, int c) {
  int q = o->qcur;
  if (o->qlen[q] >= ORC_MAX_QUEUE) abort();
  Action* act = &o->queue[q][o->qlen[q]++];
  act->kind = kind; act->piece = piece; act->a = a; act->b = b; act->c = c;
}
void eng_set_state(Oracle* o, int piece, int state) {
  eng_queue(o, ACT_SET_STATE, piece, state, 0, 0);
}
void eng_turn(Oracle* o, int piece, int q) { eng_queue(o, ACT_TURN, piece, q, 0, 0); }
void eng_move_rel(Oracle* o, int piece, int d) { eng_queue(o, ACT_MOVE_REL, piece, d, 0, 0); }
void eng_move_abs(Oracle* o, int piece, int d) { eng_queue(o, ACT_MOVE_ABS, piece, d, 0, 0); }
void eng_set_orientation(Oracle* o, int piece, int d) { eng_queue(o, ACT_SET_ORIENT, piece, d, 0, 0); }
void eng_teleport(Oracle* o, int piece, int x, int y) { eng_queue(o, ACT_TELEPORT, piece, x, y, 0); }
void eng_hit_beam(Oracle* o, int piece, int hit, int length, int radius) {
  eng_queue(o, ACT_BEAM, piece, hit, length, radius);
}
void eng_teleport_to_group(Oracle* o, int piece, uint32_t group_mask, int state,
                           int orient_mode, int rng_stream, int rng_index) {
  /* pack (mode, stream, index) into c */
  eng_queue(o, ACT_TELEPORT_GROUP, piece, (int)group_mask, state,
            orient_mode | (rng_stream << 4) | (rng_index << 12));
}

/* grid:createPiece(state, transform) — immediate, not queued
 * (component_library.lua:236-254, avatar_library.lua:305-311). */
int eng_create_piece(Oracle* o, int state, int x, int y, int orient, int kind,
                     int index) {
  int id = o->npieces++;
  Piece* p = &o->pieces[id];
  p->state = state; p->x = x; p->y = y; p->orient = orient;
  p->change_frame = o->frame; p->kind = kind; p->index = index; p->leader = -1;
  int layer = o->state_layer[state];
  if (layer >= 0) {
    int ci = cell_index(o, layer, x, y);
    if (o->cell[ci] >= 0) abort(); /* "Failed to create piece" assert */
    o->cell[ci] = id;
  }
  return id;
}

/* Placing a piece whose state has a contact name triggers
 * onContact[contact].enter on every other piece in the cell
 * (docs/advanced.md:45-49; clean_up/components.lua:390-408). */
static void fire_enter(Oracle* o, int piece) {
  const Piece* p = &o->pieces[piece];
  int contact = o->state_contact[p->state];
  if (contact < 0 || o->sub->on_enter == 0) return;
  for (int l = 0; l < o->L; ++l) {
    int other = o->cell[cell_index(o, l, p->x, p->y)];
    if (other >= 0 && other != piece) o->sub->on_enter(o, other, piece, contact);
  }
}

static int wrap_or_reject(const Oracle* o, int* x, int* y) {
  if (o->topology == 1) { /* TORUS */
    *x = ((*x % o->W) + o->W) % o->W;
    *y = ((*y % o->H) + o->H) % o->H;
    return 1;
  }
  return *x >= 0 && *x < o->W && *y >= 0 && *y < o->H;
}

/* Lift / change / place (docs/advanced.md:45-50).  Returns 1 on success. */
static int place_state(Oracle* o, int piece, int new_state, int nx, int ny) {
  Piece* p = &o->pieces[piece];
  int old_state = p->state;
  if (new_state == old_state && nx == p->x && ny == p->y) return 0; /* A2b */
  int old_layer = o->state_layer[old_state];
  int new_layer = o->state_layer[new_state];
  int lifted = 0;
  if (new_layer >= 0) {
    int occ = o->cell[cell_index(o, new_layer, nx, ny)];
    if (occ >= 0 && occ != piece) {
      if (!o->opt_set_state_lifts) return 0; /* blocked: state unchanged */
      lifted = 1;   /* A19: the state changes, the piece stays off the grid until it is moved */
    }
  }
  if (old_layer >= 0 && o->cell[cell_index(o, old_layer, p->x, p->y)] == piece)
    o->cell[cell_index(o, old_layer, p->x, p->y)] = -1;
  p->state = new_state; p->x = nx; p->y = ny;
  p->change_frame = o->frame;
  if (new_layer >= 0 && !lifted) o->cell[cell_index(o, new_layer, nx, ny)] = piece;
  /* A21: the new state's onAdd, then the contact callbacks of the cell it was placed on (the
   * order matters in externality_mushrooms only: Avatar:onStateChange restarts the freeze
   * counter a mushroom eaten on the spawn point sets, avatar_library.lua:436-437) */
  if (o->sub->on_state_change) o->sub->on_state_change(o, piece, old_state);
  if (new_layer >= 0 && !lifted && o->pieces[piece].state == new_state) fire_enter(o, piece);
  return 1;
}

/* moveAbs: "If there is a piece in the target location at the time of the move
 * then the piece stays where it is ... Both callbacks are triggered even if the
 * move is not possible and `piece` leaves and enters the same cell."
 * (component_library.lua:292-309; KATs piece_movement_test.lua:69-78,
 * game_object_test.lua:267-293). */
/* grid:connect (avatar_library.lua:388-404 "if one object is pushed or turned,
 * then they are all pushed").  A14: connected pieces move as a unit — the move
 * succeeds only if every on-grid member's target is free. */
/* events:add(name, 'dict', ...) — recorded for api:events (api_factory.lua);
 * cleared at the start of every reset / advance. */
void eng_event(Oracle* o, int type, int a, int b) {
  if (o->ev_count < ORC_MAX_EVENTS) {
    o->ev[o->ev_count][0] = type; o->ev[o->ev_count][1] = a; o->ev[o->ev_count][2] = b;
  }
  o->ev_count++;
}

void eng_connect(Oracle* o, int leader, int follower) { o->pieces[follower].leader = leader; }
/* game_object:disconnect (avatar_library.lua:400-406) — immediate, like connect */
void eng_disconnect(Oracle* o, int follower) { o->pieces[follower].leader = -1; }

static void do_move(Oracle* o, int piece, int absdir) {
  Piece* p = &o->pieces[piece];
  int layer = o->state_layer[p->state];
  if (layer < 0) return; /* off-grid pieces have no position to move from */
  if (o->cell[cell_index(o, layer, p->x, p->y)] != piece) return;   /* (A19: lifted) */
  int group[8], ng = 0;
  group[ng++] = piece;
  for (int q = 0; q < o->npieces && ng < 8; ++q)
    if (o->pieces[q].leader == piece && o->state_layer[o->pieces[q].state] >= 0 &&
        o->cell[cell_index(o, o->state_layer[o->pieces[q].state], o->pieces[q].x, o->pieces[q].y)] == q)
      group[ng++] = q;
  int ok = 1;
  for (int g = 0; g < ng; ++g) {
    const Piece* m = &o->pieces[group[g]];
    int nx = m->x + kDx[absdir], ny = m->y + kDy[absdir];
    if (!wrap_or_reject(o, &nx, &ny)) { ok = 0; break; }
    if (o->cell[cell_index(o, o->state_layer[m->state], nx, ny)] >= 0) { ok = 0; break; }
  }
  if (!ok) {
    if (o->opt_blocked_move_reenters) fire_enter(o, piece); /* A3b */
    return;
  }
  for (int g = 0; g < ng; ++g) {
    Piece* m = &o->pieces[group[g]];
    int ml = o->state_layer[m->state];
    int nx = m->x + kDx[absdir], ny = m->y + kDy[absdir];
    wrap_or_reject(o, &nx, &ny);
    o->cell[cell_index(o, ml, m->x, m->y)] = -1;
    m->x = nx; m->y = ny;
    o->cell[cell_index(o, ml, nx, ny)] = group[g];
  }
  fire_enter(o, piece);
}

static void do_teleport(Oracle* o, int piece, int x, int y) {
  Piece* p = &o->pieces[piece];
  int layer = o->state_layer[p->state];
  if (layer < 0) { p->x = x; p->y = y; return; }
  if (!wrap_or_reject(o, &x, &y)) return;
  int occ = o->cell[cell_index(o, layer, x, y)];
  if (occ >= 0 && occ != piece) {
    if (o->opt_blocked_move_reenters) fire_enter(o, piece);
    return;
  }
  if (o->cell[cell_index(o, layer, p->x, p->y)] == piece) o->cell[cell_index(o, layer, p->x, p->y)] = -1;
  p->x = x; p->y = y;
  o->cell[cell_index(o, layer, x, y)] = piece;
  fire_enter(o, piece);
}

/* teleportToGroup(piece, group, state, orient): "Sets position of GameObject to
 * any position matching any piece in a group. Calls the same add/remove
 * callbacks as grid::setState()" (component_library.lua:336-354; KAT
 * game_object_test.lua:311-345).  A5: uniform over the group's pieces in
 * creation order; an occupied target makes the teleport fail (state unchanged,
 * so a state-gated updater simply retries on the next frame). */
static void do_teleport_group(Oracle* o, const Action* a) {
  uint32_t mask = (uint32_t)a->a;
  int mode = a->c & 15, stream = (a->c >> 4) & 255, index = a->c >> 12;
  int n = 0;
  const int tl = o->state_layer[a->b];
#define TELEPORT_CANDIDATE(i)                                                    \
  ((i) != a->piece && (o->state_groups[o->pieces[i].state] & mask) &&           \
   (!o->opt_teleport_free_only || tl < 0 ||                                     \
    o->cell[cell_index(o, tl, o->pieces[i].x, o->pieces[i].y)] < 0))
  for (int i = 0; i < o->npieces; ++i)
    if (TELEPORT_CANDIDATE(i)) ++n;
  if (n == 0) return;
  PhiloxOut d = eng_draw(o, stream, (uint32_t)index);
  int k = (int)eng_bounded(o, d, (uint32_t)n), target = -1;
  for (int i = 0; i < o->npieces; ++i)
    if (TELEPORT_CANDIDATE(i))
      if (k-- == 0) { target = i; break; }
#undef TELEPORT_CANDIDATE
  const Piece* t = &o->pieces[target];
  if (!place_state(o, a->piece, a->b, t->x, t->y)) return;
  Piece* p = &o->pieces[a->piece];
  if (mode == TELEPORT_PICK_RANDOM) p->orient = (int)eng_pick4(o, d);
  else if (mode == TELEPORT_MATCH_TARGET) p->orient = t->orient;
}

/* One beam cell: every piece in the cell whose state handles the hit gets
 * onHit; any `true` stops the beam (game_object.lua:287-296).  A4: the beam
 * sprite is drawn on the hit's layer for this frame, blocked cell included. */
static int hit_cell_dir(Oracle* o, int piece, int hit, int x, int y, int dir) {
  int blocked = 0;
  for (int l = 0; l < o->L; ++l) {
    int other = o->cell[cell_index(o, l, x, y)];
    if (other >= 0 && other != piece && o->sub->on_hit)
      if (o->sub->on_hit(o, other, piece, hit)) blocked = 1;
  }
  if (!blocked || o->opt_beam_marks_blocked) {
    int hs = o->hit_state_dir ? o->hit_state_dir[hit * 4 + dir] : o->hit_state[hit];
    o->beam[cell_index(o, o->state_layer[hs], x, y)] = (uint8_t)hs;
  }
  return blocked;
}

static void ray(Oracle* o, int piece, int hit, int x, int y, int dir, int len) {
  for (int i = 1; i <= len; ++i) {
    int cx = x + i * kDx[dir], cy = y + i * kDy[dir];
    if (!wrap_or_reject(o, &cx, &cy)) return;
    if (hit_cell_dir(o, piece, hit, cx, cy, dir)) return;
  }
}

/* hitBeam(piece, hit, length, radius) (game_object.lua:246-258).  A4 footprint,
 * the one the reference itself assumes in Zapper:getWhoZappable
 * (avatar_library.lua:780-824): a centre ray of `length`; on each side walk
 * outwards up to `radius` cells (stopping at a blocker) and from each such
 * cell send a forward ray of length - offset. */
static void do_beam(Oracle* o, const Action* a) {
  const Piece* p = &o->pieces[a->piece];
  if (o->state_layer[p->state] < 0) return;
  int hit = a->a, length = a->b, radius = a->c;
  int fwd = p->orient;
  ray(o, a->piece, hit, p->x, p->y, fwd, length);
  for (int s = 0; s < 2; ++s) {
    int side = (fwd + (s == 0 ? 3 : 1)) & 3; /* left first, then right */
    for (int i = 1; i <= radius; ++i) {
      int cx = p->x + i * kDx[side], cy = p->y + i * kDy[side];
      if (!wrap_or_reject(o, &cx, &cy)) break;
      if (hit_cell_dir(o, a->piece, hit, cx, cy, fwd)) break;
      ray(o, a->piece, hit, cx, cy, fwd, length - i);
    }
  }
}

static void process(Oracle* o, const Action* a) {
  Piece* p = &o->pieces[a->piece];
  switch (a->kind) {
    case ACT_SET_STATE: place_state(o, a->piece, a->a, p->x, p->y); break;
    /* turn(angle): 1 = 90 deg clockwise, 3 = counter-clockwise
     * (component_library.lua:356-366; KAT game_object_test.lua:347-362). */
    case ACT_TURN: p->orient = (p->orient + a->a + 4) & 3; break;
    case ACT_SET_ORIENT: p->orient = a->a & 3; break;
    /* moveRel('E') while facing S moves to x-1
     * (KAT game_object_test.lua:281-293): relative E = the piece's right. */
    case ACT_MOVE_REL: do_move(o, a->piece, (p->orient + a->a) & 3); break;
    case ACT_MOVE_ABS: do_move(o, a->piece, a->a & 3); break;
    case ACT_TELEPORT: do_teleport(o, a->piece, a->a, a->b); break;
    case ACT_TELEPORT_GROUP: do_teleport_group(o, a); break;
    case ACT_BEAM: do_beam(o, a); break;
  }
}

/* grid:update(random) — dmlab2d Grid::DoUpdate (docs/advanced.md:33-52):
 * beam sprites of the previous frame disappear; updaters run in priority
 * order; queued events are processed in FIFO order; events queued by callbacks
 * are processed in the next flush of the same update (A2: flush count 128, the
 * engine's default; docs/advanced.md:51 words this as "a future update"). */
void eng_do_update(Oracle* o) {
  memset(o->beam, 0, (size_t)o->L * o->H * o->W);
  o->trace_n = 0;
  o->sub->run_updaters(o);
  for (int f = 0; f < o->opt_flush_count; ++f) {
    int cur = o->qcur;
    if (o->qlen[cur] == 0) break;
    o->qcur = cur ^ 1;
    o->qlen[o->qcur] = 0;
    for (int i = 0; i < o->qlen[cur]; ++i) process(o, &o->queue[cur][i]);
    o->qlen[cur] = 0;
  }
  o->frame++;
}
