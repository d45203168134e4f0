/* ORACLE — test infrastructure only.  See engine.h.
 *
 * Layer views + tile renderer restatement (dmlab2d `world:createView`,
 * `tile.Scene:render`; reference call sites avatar_library.lua:225-277,
 * base_simulation.lua:347-368).
 *
 *  - egocentric window left/right/forward/backward around the avatar, rotated
 *    so that the avatar faces up (A6); cells outside the map show the
 *    `OutOfBounds` sprite (base_simulation.lua:322-324); an avatar that is
 *    off-grid (wait state) sees only `OutOfBounds` ("receiving black frames",
 *    avatar_library.lua:61-62);
 *  - per-viewer sprite remap (spriteMap, avatar_library.lua:236;
 *    clean_up.py:630-631);
 *  - each sprite has 4 facings; the facing shown is the piece's orientation
 *    relative to the viewer (A9);
 *  - layers composited bottom -> top in renderOrder with 8-bit alpha (A7):
 *    out = (src*a + dst*(255-a) + 127) / 255.
 */
#include <string.h>

#include "engine.h"

enum { SPRITE_OUT_OF_BOUNDS = 0, SPRITE_OUT_OF_VIEW = 1 };

static inline const uint8_t* sprite_px(const Oracle* o, int sprite, int facing,
                                       int py, int px) {
  return o->sprite_rgba + ((((size_t)sprite * 4 + facing) * 8 + py) * 8 + px) * 4;
}

static inline void blend(uint8_t* dst, const uint8_t* src) {
  unsigned a = src[3];
  if (a == 0) return;
  for (int c = 0; c < 3; ++c)
    dst[c] = (uint8_t)((src[c] * a + dst[c] * (255u - a) + 127u) / 255u);
}

/* Composite one map cell (or OutOfBounds when x<0) into an 8x8 block. */
static void render_cell(const Oracle* o, int viewer_row, int viewer_orient,
                        int x, int y, int in_bounds, uint8_t* rgb, int stride) {
  const int32_t* remap = o->view_sprite_map + (size_t)viewer_row * o->nsprites;
  for (int py = 0; py < 8; ++py) memset(rgb + py * stride, 0, 24);
  if (!in_bounds) {
    for (int py = 0; py < 8; ++py)
      for (int px = 0; px < 8; ++px)
        blend(rgb + py * stride + px * 3,
              sprite_px(o, remap[SPRITE_OUT_OF_BOUNDS], 0, py, px));
    return;
  }
  for (int l = 0; l < o->L; ++l) {
    int idx = (l * o->H + y) * o->W + x;
    int piece = o->cell[idx];
    int state, orient = 0;
    if (piece >= 0) { state = o->pieces[piece].state; orient = o->pieces[piece].orient; }
    else if (o->beam[idx]) {
      state = o->beam[idx];
      if (o->state_orient) orient = o->state_orient[state]; /* oriented beam sprite */
    }
    else continue;
    int sprite = o->state_sprite[state];
    if (sprite < 0) continue;
    sprite = remap[sprite];
    int facing = (orient - viewer_orient) & 3;
    for (int py = 0; py < 8; ++py)
      for (int px = 0; px < 8; ++px)
        blend(rgb + py * stride + px * 3, sprite_px(o, sprite, facing, py, px));
  }
}

void orc_render_world(const Oracle* o, uint8_t* rgb) {
  int stride = o->W * 8 * 3;
  for (int y = 0; y < o->H; ++y)
    for (int x = 0; x < o->W; ++x)
      render_cell(o, o->P_pack, 0, x, y, 1, rgb + (size_t)y * 8 * stride + x * 24,
                  stride);
}

void orc_render_view(const Oracle* o, int player, uint8_t* rgb) {
  const int vl = o->hdr[10], vr = o->hdr[11], vf = o->hdr[12], vb = o->hdr[13];
  const int vw = vl + vr + 1, vh = vf + vb + 1;
  const int stride = vw * 8 * 3;
  const Piece* p = &o->pieces[o->avatar_piece[player]];
  int on_grid = o->state_layer[p->state] >= 0;
  for (int vy = 0; vy < vh; ++vy)
    for (int vx = 0; vx < vw; ++vx) {
      uint8_t* dst = rgb + (size_t)vy * 8 * stride + vx * 24;
      if (!on_grid && o->opt_dead_view_black) {
        render_cell(o, player, 0, -1, -1, 0, dst, stride);
        continue;
      }
      int dx = vx - vl, dy = vy - vf; /* right, down in the viewer's frame */
      int ax, ay;
      switch (p->orient) {
        case ORIENT_N: ax = dx; ay = dy; break;
        case ORIENT_E: ax = -dy; ay = dx; break;
        case ORIENT_S: ax = -dx; ay = -dy; break;
        default: ax = dy; ay = -dx; break;
      }
      int x = p->x + ax, y = p->y + ay, in = 1;
      if (o->topology == 1) {
        x = ((x % o->W) + o->W) % o->W;
        y = ((y % o->H) + o->H) % o->H;
      } else {
        in = x >= 0 && x < o->W && y >= 0 && y < o->H;
      }
      render_cell(o, player, p->orient, x, y, in, dst, stride);
    }
}

/* "N.LAYER" (avatar_library.lua:246-257): the player's layer view with
 * `orientation = 'N'` — the window is NOT turned with the avatar — as int32
 * [vh][vw][L].  A17 (nothing in the reference pins dmlab2d's sprite ids): a cell
 * of a layer holds 1 + the index of the sprite of the piece (or beam) seen
 * there, after the viewer's spriteMap (sprites in the order the level registers
 * them = the pack's sprite_names); 0 = nothing; every layer of a cell outside
 * the map — and of every cell of an off-grid viewer (A6) — holds the
 * OutOfBounds sprite. */
void orc_layer_view(const Oracle* o, int player, int32_t* out) {
  const int vl = o->hdr[10], vr = o->hdr[11], vf = o->hdr[12], vb = o->hdr[13];
  const int vw = vl + vr + 1, vh = vf + vb + 1;
  const Piece* p = &o->pieces[o->avatar_piece[player]];
  const int32_t* remap = o->view_sprite_map + (size_t)player * o->nsprites;
  int on_grid = o->state_layer[p->state] >= 0;
  for (int vy = 0; vy < vh; ++vy)
    for (int vx = 0; vx < vw; ++vx) {
      int32_t* dst = out + ((size_t)vy * vw + vx) * o->L;
      int x = p->x + (vx - vl), y = p->y + (vy - vf), in = 1;
      if (o->topology == 1) {
        x = ((x % o->W) + o->W) % o->W;
        y = ((y % o->H) + o->H) % o->H;
      } else {
        in = x >= 0 && x < o->W && y >= 0 && y < o->H;
      }
      if (!in || (!on_grid && o->opt_dead_view_black)) {
        for (int l = 0; l < o->L; ++l) dst[l] = 1 + remap[SPRITE_OUT_OF_BOUNDS];
        continue;
      }
      for (int l = 0; l < o->L; ++l) {
        int idx = (l * o->H + y) * o->W + x;
        int piece = o->cell[idx], state;
        dst[l] = 0;
        if (piece >= 0) state = o->pieces[piece].state;
        else if (o->beam[idx]) state = o->beam[idx];
        else continue;
        int sprite = o->state_sprite[state];
        if (sprite < 0) continue;
        dst[l] = 1 + remap[sprite];
      }
    }
}
