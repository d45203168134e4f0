// step_cook.h — one environment step (or episode start) of one collaborative_cooking world
// by one wavefront (shape: step_clean_up.h).
//
// Substrate rules restated here (reference: configs/substrates/collaborative_cooking.py + the
// seven layout modules, lua/levels/collaborative_cooking/components.lua):
//   InteractBeam  :29-113   cooldown run down in the updater (priority 140); a beam one cell
//                           long, its own hit / layer / sprite per avatar
//   Container     :116-181  counters and dispensers: the avatar and the inventory piece over
//                           the container swap an item when exactly one of them holds one (a
//                           dispenser keeps its own); once per frame (`_usedThisStep`)
//   Inventory     :184-277  a piece whose state IS the item; an avatar's is connected to it
//                           (moves with it, A14; turned with it, avatar_library.lua:162-164)
//   Receiver      :280-333  takes the accepted item, pays the avatar — or every avatar
//   CookingPot    :336-474  up to three ingredients (a Lua list: a count here), cooks for
//                           `cookingTime` ticks once full, hands the soup to a dish
//   LoadingBarVisualiser :477-517  the bar over a pot shows the pot's time as the PREVIOUS
//                           frame's tick left it (its updater is registered first: A11)
// getHeldItem reads a piece's state as the engine has it: a setState queued by an earlier hit
// of the same flush is not seen.  Nothing in a frame reads a state another hit of the frame
// queued (an avatar fires once; a container answers once), so the queued states are written
// where they are decided; a pot's content, time and `cooked` are Lua variables (one byte of a
// hidden plane: time in bits 0-4, content in 5-6, cooked in 7) and change at once.
// An avatar's inventory shows its facing in its state id (the pack's pseudo-states per
// (item, facing): this engine keeps no orientation of non-avatar pieces).
#ifndef MP_STEP_COOK_H_
#define MP_STEP_COOK_H_

#include "step_common.h"

namespace stepk {

constexpr int kContRegs = 2;   // mp_create admits at most 128 containers and 64 pots

struct CookSites { int cont[kContRegs]; int start[kContRegs]; int pot; };

__device__ inline CookSites load_sites(const CookTables& c, int lane) {
  CookSites s;
#pragma unroll
  for (int k = 0; k < kContRegs; ++k) {
    const int i = k * 64 + lane;
    s.cont[k] = i < c.n_cont ? c.cont_cells[i] : -1;
    s.start[k] = i < c.n_cont ? c.cont_i32[2 * i] : 0;
  }
  s.pot = lane < c.n_pot ? c.pot_cells[lane] : -1;
  return s;
}

enum { COOK_EMPTY = 0, COOK_TOMATO = 1, COOK_DISH = 2, COOK_SOUP = 3 };
enum { COOK_KIND_CONTAINER = 1, COOK_KIND_DISPENSER = 2, COOK_KIND_RECEIVER = 3, COOK_KIND_POT = 4 };

// Inventory:getHeldItem of a state id (-1: not an inventory state, e.g. 'wait' or nothing)
__device__ inline int cook_item_of(const CookTables& c, int s) {
  if (s >= c.s_plain0 && s < c.s_plain0 + 4) return s - c.s_plain0;
  if (s >= c.s_off0 && s < c.s_off0 + 4) return s - c.s_off0;
  if (s >= c.s_dir0 && s < c.s_dir0 + 12) return (s - c.s_dir0) & 3;
  return -1;
}
// the state of an avatar's inventory holding `item`, facing `ori`
__device__ inline int cook_offset_state(const CookTables& c, int item, int ori) {
  return ori == 0 ? c.s_off0 + item : c.s_dir0 + (ori - 1) * 4 + item;
}

__device__ inline void step_world(const DevTables& t, const CookTables& c,
                                  const CookSites& sites, const World& wd, const Action& act,
                                  const StepArgs& args) {
  const int lane = wd.lane, w = wd.w;
  const StepOutputs& out = args.out;
  Scratch* sc = wd.sc;
  uint8_t* grid = wd.rec;
  WorldTail* tail = reinterpret_cast<WorldTail*>(wd.rec + t.grid_pad);
  const int P = t.P, HW = t.H * t.W, W = t.W;
  const bool is_av = lane < P;
  auto at = [&](int layer, int cell) -> uint8_t& { return grid[layer * HW + cell]; };

  const OrderStreams kOrders = {RS_SHUFFLE_MOVE, RS_SHUFFLE_ZAP, 0, 0, 2};   // the updater groups shuffled per frame (A1)

  const int what = dispatch(t, tail, lane, w, args.reset_mask, args.mode, args.auto_reset, out);
  if (what == 0) return;

  Av a;
  int step_type;
  const int alive_state = is_av ? t.alive_state[lane] : 0;
  const int ov = c.overlay_layer;

  if (what == 1) {
    // ---- api:start (api_factory.lua:85-102); the episode number is a word of the draw
    // counter (A10; the reference re-seeds with seed + 1, builder.py:177-181)
    const uint32_t k0 = (uint32_t)tail->seed, k1 = (uint32_t)(tail->seed >> 32);
    const uint32_t ep = tail->episode;
    wsync();
    const int gvec = (t.L * HW + 15) >> 4;
    for (int i = lane; i < gvec; i += 64)
      reinterpret_cast<uint4*>(grid)[i] = reinterpret_cast<const uint4*>(t.init_grid)[i];
    wsync();
    for (int i = lane; i < HW; i += 64) at(c.plane_t, i) = 0;   // CookingPot:reset
    if (lane == 0) {
      tail->episode = ep + 1;
      tail->step = 0; tail->frame = 1; tail->done = 0; tail->cont = 1;
      tail->started = 1;
      tail->aux_count = 0;
      tail->group_change = 0;
      tail->ctr[2]++;
    }
    wsync();
    apply_map_choices(t, grid, lane, ep, k0, k1);
    spawn_avatars(t, grid, lane, ep, k0, k1, a);
    if (lane < P) push_event(sc, MP_EVENT_AVATAR_STARTED, 0, 0);
    // Inventory:postStart (components.lua:224-243): the inventory over a container holds the
    // container's startingItem; an avatar's sits on its avatar in the plain 'empty' state
#pragma unroll
    for (int k = 0; k < kContRegs; ++k)
      if (sites.cont[k] >= 0) at(ov, sites.cont[k]) = (uint8_t)(c.s_plain0 + sites.start[k]);
    if (is_av) at(ov, a.y * W + a.x) = (uint8_t)(c.s_plain0 + COOK_EMPTY);
    step_type = 0;
  } else {
    // ================= api:advance =================
    const uint32_t k0 = (uint32_t)tail->seed, k1 = (uint32_t)(tail->seed >> 32);
    const uint32_t ep = tail->episode - 1;
    const int step = tail->step + 1, frame = tail->frame;
    load_avatars(tail, lane, a);
    wsync();
    const int a_move = act.move, a_turn = act.turn, a_interact = act.fire0, bad = act.bad;

    // ---- updaters (pre-flush state)
    int orders[4];
    step_orders(tail, lane, P, kOrders, (uint32_t)step, ep, k0, k1, orders);
    const int order_move = orders[0], order_interact = orders[1];
    // 140 LoadingBarVisualiser (registered first): the pot's time before this frame's tick
    if (sites.pot >= 0) {
      const int idx = (at(c.plane_t, sites.pot) & 31) / c.bar_interval;
      at(ov, sites.pot) = (uint8_t)(c.s_bar0 + (idx < 10 ? idx : 10));
    }
    // 140 InteractBeam (components.lua:79-100)
    bool fire = false;
    if (is_av && c.cooldown >= 0) {
      if (a.ztimer > 0) a.ztimer--;
      else if (a_interact == 1) { a.ztimer = c.cooldown; fire = true; }
    }
    // 140 CookingPot tickPotFn (:452-470); its setState(cooked) is queued BEHIND the beams
    bool cooked_now = false;
    if (sites.pot >= 0) {
      int v = at(c.plane_t, sites.pot);
      if (((v >> 5) & 3) == 3 && !(v & 128)) {
        if ((v & 31) == c.cooking_time) { v |= 128; cooked_now = true; }
        v = (v & ~31) | (((v & 31) + 1) & 31);
        at(c.plane_t, sites.pot) = (uint8_t)v;
      }
    }
    // beam sprites of the previous frame disappear (grid:update start)
    for (int p = 0; p < P; ++p) clear_bytes(grid, (c.beam_layer0 + p) * HW, HW, lane);
    wsync();

    // ---- flush 1: the moves, in visiting order; the inventory goes with its avatar (A14)
    // and turns with it
    const int held_state = is_av ? at(ov, a.y * W + a.x) : 0;
    (void)resolve_moves(t, wd, a, a_move, a_turn, order_move, alive_state, ov);
    int item = cook_item_of(c, held_state);
    const bool plain = held_state >= c.s_plain0 && held_state < c.s_plain0 + 4;   // (before its first setHeldItem)
    if (is_av && !plain && item >= 0) at(ov, a.y * W + a.x) = (uint8_t)cook_offset_state(c, item, a.ori);
    wsync();

    // ... then the beams, one at a time in the interact updater's visiting order
    int queued = 0;                       // this lane's pot got a setState from a hit
    bool changed = false;                 // this lane's avatar's setHeldItem
    if (__ballot(fire) != 0ull) {
      const int firing = (int)fire;
      for (int r = 0; r < P; ++r) {
        const int g = rdlane(order_interact, r);
        if (rdlane(firing, g) == 0) continue;
        int tx = rdlane(a.x, g), ty = rdlane(a.y, g);
        const int dir = rdlane(a.ori, g);
        if (!step_cell(t, tx, ty, dir_dx(dir), dir_dy(dir))) continue;    // off the map: no cell, no sprite
        const int cell = ty * W + tx;
        if (lane == 0) at(c.beam_layer0 + g, cell) = (uint8_t)(c.s_beam0 + g);   // A4
        const int s = at(t.avatar_layer, cell);
        const int kind = s ? c.state_kind[s] : 0;
        const int mine = rdlane(item, g);
        if (kind == COOK_KIND_CONTAINER || kind == COOK_KIND_DISPENSER) {
          // Container:onHit (:137-163)
          if (wd.mark[cell] == 0) {
            const int its = cook_item_of(c, at(ov, cell));
            wsync();
            if (lane == 0) wd.mark[cell] = 1;
            if (its > COOK_EMPTY && mine == COOK_EMPTY) {
              if (lane == g) { item = its; changed = true; }
              if (kind == COOK_KIND_CONTAINER && lane == 0) at(ov, cell) = (uint8_t)(c.s_plain0 + COOK_EMPTY);
            } else if (its == COOK_EMPTY && mine > COOK_EMPTY) {
              if (lane == g) { item = COOK_EMPTY; changed = true; }
              if (lane == 0) at(ov, cell) = (uint8_t)(c.s_plain0 + mine);
            }
          }
        } else if (kind == COOK_KIND_RECEIVER) {
          // Receiver:onHit (:301-333)
          if (mine == c.recv_item) {
            if (c.recv_global) { if (is_av && a.alive) a.reward += c.recv_reward; }
            else if (lane == g) a.reward += c.recv_reward;
            if (lane == g) {
              item = COOK_EMPTY; changed = true;
              push_event(sc, MP_EVENT_RECEIVER_ACCEPTED_ITEM, g + 1, mine);
            }
          }
        } else if (kind == COOK_KIND_POT) {
          // CookingPot:onHit (:378-448)
          int v = at(c.plane_t, cell);
          const int count = (v >> 5) & 3;
          if (mine == COOK_TOMATO && count < 3) {
            v += 32;
            if (lane == g) {
              a.reward += c.pot_reward; item = COOK_EMPTY; changed = true;
              push_event(sc, MP_EVENT_ITEM_DROPPED_INTO_POT, g + 1, mine);
            }
          } else if (mine == COOK_DISH && (v & 128)) {
            v = 0;
            if (lane == g) {
              a.reward += c.pot_reward; item = COOK_SOUP; changed = true;
              push_event(sc, MP_EVENT_COOKED_FOOD_COLLECTED, g + 1, COOK_SOUP);
            }
          }
          wsync();
          if (lane == 0) at(c.plane_t, cell) = (uint8_t)v;
          if (!(v & 128) && sites.pot == cell) queued = 1;
        }
        wsync();   // the next hit reads what this one wrote
      }
    }
    // ... then the pots' own setState(cooked), and — flush 2 — what the hits queued: the
    // last state a hit queued for a pot is the one of its final content
    if (sites.pot >= 0) {
      const int v = at(c.plane_t, sites.pot);
      const int k = (v >> 5) & 3;
      if (queued) at(t.avatar_layer, sites.pot) = (uint8_t)(k == 0 ? c.s_pot[0] : k == 1 ? c.s_pot[1] : k == 2 ? c.s_pot[2] : c.s_pot[3]);
      else if (cooked_now) at(t.avatar_layer, sites.pot) = (uint8_t)c.s_pot[4];
    }
    if (is_av && changed) at(ov, a.y * W + a.x) = (uint8_t)cook_offset_state(c, item, a.ori);
    // Container tick: `_usedThisStep` starts the next frame cleared
#pragma unroll
    for (int k = 0; k < kContRegs; ++k)
      if (sites.cont[k] >= 0) wd.mark[sites.cont[k]] = 0;
    wsync();
    const unsigned long long badb = __ballot(bad != 0);
    const int done = !(tail->cont && step < t.max_frames);
    if (lane == 0) {
      tail->step = step;
      tail->frame = frame + 1;
      tail->done = done;
      tail->ctr[0]++; tail->ctr[1] += (uint32_t)P; tail->ctr[7] += __popcll(badb);
    }
    step_type = done ? 2 : 1;
  }
  finish(t, wd, tail, a, 0.0, c.cooldown > 0 ? c.cooldown : 1, step_type, out, kOrders);
}

}  // namespace stepk

#endif  // MP_STEP_COOK_H_
