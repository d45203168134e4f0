// step_clean_up.hip — one environment step (or episode start) of N clean_up
// worlds, one wavefront per world.
//
// Replaces, for the clean_up substrate, the reference's per-step path
//   api:advance            lua/modules/api_factory.lua:104-111
//   BaseSimulation:update  lua/modules/base_simulation.lua:476-486
//   grid:update(random)    dmlab2d (documented cycle: docs/advanced.md:33-52)
// and the Lua component callbacks it drives (cited at each block below), and
// the episode start path api:start (api_factory.lua:85-102,
// base_simulation.lua:396-471).
//
// Execution shape: a 64-lane workgroup streams its world record (grid planes +
// tail, ≈6 KB) from HBM into LDS with 16-byte lane loads, steps it there, and
// streams it back.  Site-parallel work (122 AppleGrow draws, 147 dirt sites,
// 167 water pieces, plane clears) runs across the lanes with ballots for the
// set-selection; the ordered event phase (moves in the frame's shuffled
// order, beams, respawns) runs on lane 0 against LDS, because the reference
// resolves those sequentially and the order is observable.
#include "mp_common.h"

namespace {

constexpr int kDx[4] = {0, 1, 0, -1};  // N E S W; N = decreasing y
constexpr int kDy[4] = {-1, 0, 1, 0};  // (component_library.lua:379-386)

enum { HIT_ZAP = 0, HIT_CLEAN = 1 };
enum { A_MOVE = 0, A_TURN = 1, A_ZAP = 2, A_CLEAN = 3 };

// Per-wave scratch placed after the world record in LDS.
struct Scratch {
  double reward[MP_MAX_PLAYERS];
  double aux0[MP_MAX_PLAYERS];
  int8_t act[MP_MAX_PLAYERS][4];
  uint8_t order[4][MP_MAX_PLAYERS];  // move, zap, clean, respawn orders
  uint8_t fire[2][MP_MAX_PLAYERS];   // queued beams (player ids), by hit
  uint8_t respawn[MP_MAX_PLAYERS];
  uint8_t n_fire[2], n_respawn, pad0;
  uint16_t pend_apple[2 * MP_MAX_PLAYERS];
  uint16_t pend_dirt[MP_MAX_PLAYERS * 16];
  int32_t n_pend_apple, n_pend_dirt;
  uint32_t zapped_mask;
  int32_t spawn_site, water_advance;
  uint8_t hit_block[256];
  int8_t splayer[256];
};

struct World {
  const DevTables& t;
  const CleanUpTables& c;
  uint8_t* grid;
  WorldTail* tail;
  Scratch* sc;
  uint32_t k0, k1;
  int HW;

  __device__ uint8_t& at(int layer, int cell) { return grid[layer * HW + cell]; }
  __device__ Philox4 draw(int stream, uint32_t index) const {
    return philox4x32_10(index, (uint32_t)stream, (uint32_t)tail->step, 0u, k0, k1);
  }
};

// A1: the engine visits the pieces of an updater group in a freshly shuffled
// order every frame; forward Fisher-Yates, one draw per position.
__device__ void shuffle_order(World& wd, int stream, uint8_t* items, int n) {
  for (int i = 0; i < n; ++i) items[i] = (uint8_t)i;
  for (int i = 0; i + 1 < n; ++i) {
    int j = i + (int)philox_bounded(wd.draw(stream, (uint32_t)i), (uint32_t)(n - i));
    uint8_t tmp = items[i]; items[i] = items[j]; items[j] = tmp;
  }
}

// Edible:onEnter (clean_up/components.lua:390-408) + Taste:consumed (:446-455)
// for an avatar placed on `cell` (onContact 'avatar' enter, docs/advanced.md:45-49).
__device__ void fire_enter(World& wd, int p, int cell) {
  if (wd.at(wd.c.apple_layer, cell) == wd.c.s_apple) {
    wd.sc->reward[p] += wd.c.eat_reward;
    wd.tail->flag1[p] = 1;  // GlobalData:setAteThisStep
    wd.sc->pend_apple[wd.sc->n_pend_apple++] = (uint16_t)cell;  // -> appleWait next flush
  }
}

__device__ bool step_cell(const DevTables& t, int& x, int& y, int dir, int n) {
  x += n * kDx[dir]; y += n * kDy[dir];
  if (t.topology == 1) {  // TORUS
    x = ((x % t.W) + t.W) % t.W; y = ((y % t.H) + t.H) % t.H;
    return true;
  }
  return x >= 0 && x < t.W && y >= 0 && y < t.H;
}

// One beam cell (game_object.lua:287-296): every piece in the cell gets onHit;
// any `true` stops the beam.  A4: the beam sprite is drawn on the hit's layer
// for this frame, blocked cell included.
__device__ bool hit_cell(World& wd, int p, int hit, int x, int y) {
  const int cell = y * wd.t.W + x;
  bool blocked = false;
  for (int l = 0; l < wd.t.L; ++l) {
    const int s = wd.at(l, cell);
    if (s == 0) continue;
    // BeamBlocker:onHit (component_library.lua:678-685)
    if (wd.sc->hit_block[s] & (1u << hit)) blocked = true;
    const int pl = wd.sc->splayer[s];
    if (pl >= 0 && hit == HIT_ZAP) {
      // Zapper:onHit (avatar_library.lua:652-681); target is on-grid => alive
      wd.sc->reward[pl] += wd.c.zap_penalty;
      wd.sc->reward[p] += wd.c.zap_reward;
      if (wd.c.remove_hit) wd.sc->zapped_mask |= 1u << pl;
      wd.tail->ctr[4]++;
      blocked = true;
    }
    if (hit == HIT_CLEAN && s == wd.c.s_dirt) {
      // DirtCleaning:onHit (clean_up/components.lua:141-157)
      wd.sc->pend_dirt[wd.sc->n_pend_dirt++] = (uint16_t)cell;
      wd.tail->flag0[p] = 1;  // GlobalData:setCleanedThisStep
      wd.tail->ctr[5]++;
      blocked = true;
    }
  }
  wd.at(hit == HIT_ZAP ? wd.c.zap_layer : wd.c.clean_layer, cell) =
      (uint8_t)(hit == HIT_ZAP ? wd.c.s_zap_hit : wd.c.s_clean_hit);
  return blocked;
}

__device__ void ray(World& wd, int p, int hit, int x, int y, int dir, int len) {
  for (int i = 1; i <= len; ++i) {
    if (!step_cell(wd.t, x, y, dir, 1)) return;
    if (hit_cell(wd, p, hit, x, y)) return;
  }
}

// hitBeam(hit, length, radius) (game_object.lua:246-258) with the footprint the
// reference assumes in Zapper:getWhoZappable (avatar_library.lua:780-824).
__device__ void beam(World& wd, int p, int hit, int length, int radius) {
  if (!wd.tail->aalive[p]) return;
  const int x = wd.tail->ax[p], y = wd.tail->ay[p], fwd = wd.tail->aori[p];
  ray(wd, p, hit, x, y, fwd, length);
  for (int s = 0; s < 2; ++s) {
    const int side = (fwd + (s == 0 ? 3 : 1)) & 3;  // left first, then right
    int cx = x, cy = y;
    for (int i = 1; i <= radius; ++i) {
      if (!step_cell(wd.t, cx, cy, side, 1)) break;
      if (hit_cell(wd, p, hit, cx, cy)) break;
      ray(wd, p, hit, cx, cy, fwd, length - i);
    }
  }
}

// Lane-parallel selection of the k-th site (ascending) with pred true.
// Returns the site index (wave-uniform).
template <class Pred>
__device__ int count_sites(int lane, int n, Pred pred, unsigned long long* masks) {
  int total = 0;
  const int chunks = (n + 63) >> 6;
  for (int ch = 0; ch < chunks; ++ch) {
    const int site = ch * 64 + lane;
    const bool v = site < n && pred(site);
    const unsigned long long m = __ballot(v);
    masks[ch] = m;
    total += __popcll(m);
  }
  return total;
}
__device__ int kth_site(const unsigned long long* masks, int chunks, int k) {
  for (int ch = 0; ch < chunks; ++ch) {
    unsigned long long m = masks[ch];
    const int pc = __popcll(m);
    if (k < pc) {
      for (int i = 0; i < k; ++i) m &= m - 1;
      return ch * 64 + __ffsll((long long)m) - 1;
    }
    k -= pc;
  }
  return -1;
}

__global__ __launch_bounds__(64) void k_step_clean_up(
    DevTables t, CleanUpTables c, uint8_t* __restrict__ state,
    const int32_t* __restrict__ actions, const uint8_t* __restrict__ reset_mask,
    int mode, int auto_reset, StepOutputs out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int w = blockIdx.x, lane = threadIdx.x;
  uint8_t* gw = state + (size_t)w * t.world_stride;
  const int nvec = t.world_stride >> 4;
  for (int i = lane; i < nvec; i += 64)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(gw)[i];
  Scratch* sc = reinterpret_cast<Scratch*>(smem + t.world_stride);
  for (int s = lane; s < 256; s += 64) {
    sc->hit_block[s] = s < t.nstates ? (uint8_t)t.state_hit_block[s] : 0;
    sc->splayer[s] = s < t.nstates ? t.state_player[s] : (int8_t)-1;
  }
  __syncthreads();
  WorldTail* tail = reinterpret_cast<WorldTail*>(smem + t.grid_pad);
  World wd{t, c, smem, tail, sc, 0u, 0u, t.H * t.W};
  const int P = t.P, HW = t.H * t.W;

  bool do_reset;
  if (mode == STEP_MODE_RESET) {
    do_reset = reset_mask ? reset_mask[w] != 0 : true;
    if (!do_reset) return;
  } else {
    if (!tail->started) return;  // never reset: nothing to step
    do_reset = tail->done && auto_reset;
    if (tail->done && !auto_reset) {  // frozen after LAST until mp_reset
      if (lane < P) {
        out.reward[w * P + lane] = 0.0;
      }
      if (lane == 0) { out.collective[w] = 0.0; out.step_type[w] = 2; out.discount[w] = 0.0; }
      return;
    }
  }

  int step_type;
  if (do_reset) {
    // ---- api:start(episode, seed) (api_factory.lua:85-102); every reset of a
    // world uses seed + #earlier resets (builder.py:177-181).
    const uint64_t seed = tail->seed + tail->episode;
    wd.k0 = (uint32_t)seed; wd.k1 = (uint32_t)(seed >> 32);
    const int gvec = t.grid_pad >> 4;
    for (int i = lane; i < gvec; i += 64)
      reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(t.init_grid)[i];
    if (lane == 0) {
      tail->episode++;
      tail->step = 0; tail->frame = 0; tail->done = 0; tail->cont = 1;
      tail->started = 1;
      tail->aux_count = c.n_dirt_init;  // DirtTracker:postStart (:103-116)
      tail->group_change = 0;
      tail->ctr[2]++;
    }
    __syncthreads();
    // _avatarStart: groupShuffledWithCount(random, spawnGroup, numAvatars)
    // (base_simulation.lua:416-421): partial Fisher-Yates over the group's
    // pieces in creation order; avatar i takes the i-th sampled point.
    if (lane == 0) {
      uint16_t* spawn = sc->pend_dirt;  // scratch reuse, n_spawn <= 256
      for (int i = 0; i < t.n_spawn; ++i) spawn[i] = (uint16_t)t.spawn_cells[i];
      for (int i = 0; i < P; ++i) {
        int j = i + (int)philox_bounded(wd.draw(RS_START_SPAWN, (uint32_t)i),
                                        (uint32_t)(t.n_spawn - i));
        uint16_t tmp = spawn[i]; spawn[i] = spawn[j]; spawn[j] = tmp;
      }
    }
    __syncthreads();
    if (lane < MP_MAX_PLAYERS) {
      const int p = lane;
      const bool live = p < P;
      int cell = live ? sc->pend_dirt[p] : 0;
      // Avatar:start (avatar_library.lua:288-320): random:choice(_COMPASS)
      int orient = live ? (int)philox_bounded(wd.draw(RS_START_ORIENT, (uint32_t)p), 4u) : 0;
      tail->ax[p] = (uint8_t)(cell % t.W); tail->ay[p] = (uint8_t)(cell / t.W);
      tail->aori[p] = (uint8_t)orient; tail->aalive[p] = live ? 1 : 0;
      tail->ztimer[p] = 0; tail->ctimer[p] = 0;  // Zapper:start, Cleaner:reset
      tail->flag0[p] = 0; tail->flag1[p] = 0;    // GlobalData:reset
      tail->achange[p] = 0;
      sc->reward[p] = 0.0; sc->aux0[p] = 0.0;
      if (live) wd.at(t.avatar_layer, cell) = (uint8_t)t.alive_state[p];
    }
    // Animation:postStart with randomStartFrame (component_library.lua:1064):
    // the queued setState is flushed by the grid:update at api_factory.lua:101.
    for (int i = lane; i < c.n_water; i += 64) {
      uint32_t k = philox_bounded(wd.draw(RS_ANIM_START, (uint32_t)i), 4u);
      wd.at(c.water_layer, c.water_cells[i]) = (uint8_t)c.s_water[k];
    }
    __syncthreads();
    if (lane == 0) tail->frame = 1;
    step_type = 0;
  } else {
    // ================= api:advance =================
    const uint64_t seed = tail->seed + (tail->episode - 1);
    wd.k0 = (uint32_t)seed; wd.k1 = (uint32_t)(seed >> 32);
    __syncthreads();
    if (lane == 0) {
      tail->step++;
      tail->ctr[0]++; tail->ctr[1] += (uint32_t)P;
      sc->n_pend_apple = 0; sc->n_pend_dirt = 0; sc->zapped_mask = 0;
      sc->spawn_site = -1;
    }
    // api:discreteActions (api_factory.lua:81) + the ACTION_SET lookup of
    // discrete_action_wrapper.py:97-109; Avatar:preUpdate resets the reward.
    if (lane < MP_MAX_PLAYERS) {
      int a = lane < P ? actions[(size_t)w * P + lane] : 0;
      if (a < 0 || a >= t.nact) { a = 0; atomicAdd(&tail->ctr[7], 1u); }
      for (int k = 0; k < 4; ++k) sc->act[lane][k] = (int8_t)t.action_table[a * 4 + k];
      sc->reward[lane] = 0.0;
    }
    // beam sprites of the previous frame disappear (grid:update start)
    for (int i = lane; i < HW; i += 64) {
      wd.at(c.zap_layer, i) = 0;
      wd.at(c.clean_layer, i) = 0;
    }
    __syncthreads();
    const int step = tail->step;
    const int frame = tail->frame;

    // ---- BaseSimulation:update: DirtSpawner:update (clean_up/components.lua:329-340)
    if (step > c.dirt_delay) {
      const Philox4 d = wd.draw(RS_DIRT_SPAWN, 0);
      if (philox_u53(d) < c.thr_dirt_spawn) {
        unsigned long long masks[4];
        const int n = count_sites(lane, c.n_dirt, [&](int site) {
          return wd.at(c.dirt_wait_layer, c.dirt_cells[site]) == c.s_dirt_wait;
        }, masks);
        if (n > 0) {  // random:choice(set.toSortedList(potential))
          const int k = (int)philox_bounded(d, (uint32_t)n);
          const int site = kth_site(masks, (c.n_dirt + 63) >> 6, k);
          if (lane == 0) sc->spawn_site = site;
        }
      }
    }
    // ---- AppleGrow:update (clean_up/components.lua:64-80): one draw per
    // potential apple; the growth probability depends on the dirt count only.
    {
      const uint64_t thr = c.apple_thr[tail->aux_count];
      __syncthreads();
      // flush 1, first events: the DirtSpawner setState, then the AppleGrow
      // setStates (queue order = object creation order: scene first).
      if (lane == 0 && sc->spawn_site >= 0) {
        const int cell = c.dirt_cells[sc->spawn_site];
        if (wd.at(c.dirt_layer, cell) == 0) {
          wd.at(c.dirt_wait_layer, cell) = 0;
          wd.at(c.dirt_layer, cell) = (uint8_t)c.s_dirt;
          tail->aux_count++;  // DirtTracker:onStateChange (:118-129)
        }
      }
      for (int i = lane; i < c.n_apple; i += 64) {
        const uint64_t u = philox_u53(wd.draw(RS_APPLE_GROW, (uint32_t)i));
        if (u < thr) {
          const int cell = c.apple_cells[i];
          if (wd.at(c.apple_layer, cell) == 0) wd.at(c.apple_layer, cell) = (uint8_t)c.s_apple;
        }
      }
    }
    __syncthreads();

    if (lane == 0) {
      // ---- updaters, priority descending (updater_registry.lua:166-173)
      // 150 Avatar move order; 140 zap; 140 clean; 135 respawn
      shuffle_order(wd, RS_SHUFFLE_MOVE, sc->order[0], P);
      shuffle_order(wd, RS_SHUFFLE_ZAP, sc->order[1], P);
      shuffle_order(wd, RS_SHUFFLE_CLEAN, sc->order[2], P);
      shuffle_order(wd, RS_SHUFFLE_RESPAWN, sc->order[3], P);
      sc->n_fire[0] = sc->n_fire[1] = 0; sc->n_respawn = 0;
      // Zapper zap updater (avatar_library.lua:613-636)
      for (int i = 0; i < P; ++i) {
        const int p = sc->order[1][i];
        if (!tail->aalive[p] || c.zap_cooldown < 0) continue;
        if (tail->ztimer[p] > 0) tail->ztimer[p]--;
        else if (sc->act[p][A_ZAP] == 1) {
          tail->ztimer[p] = (uint8_t)c.zap_cooldown;
          sc->fire[0][sc->n_fire[0]++] = (uint8_t)p;
        }
      }
      // Cleaner clean updater (clean_up/components.lua:201-224)
      for (int i = 0; i < P; ++i) {
        const int p = sc->order[2][i];
        if (!tail->aalive[p] || c.clean_cooldown < 0) continue;
        if (tail->ctimer[p] > 0) tail->ctimer[p]--;
        else if (sc->act[p][A_CLEAN] == 1) {
          tail->ctimer[p] = (uint8_t)c.clean_cooldown;
          sc->fire[1][sc->n_fire[1]++] = (uint8_t)p;
        }
      }
      // Zapper respawn updater: state = waitState, startFrame = framesTillRespawn
      // (avatar_library.lua:638-649)
      for (int i = 0; i < P; ++i) {
        const int p = sc->order[3][i];
        if (tail->aalive[p]) continue;
        if (frame - tail->achange[p] < c.respawn_frames) continue;
        sc->respawn[sc->n_respawn++] = (uint8_t)p;
      }
      // 100 Animation (component_library.lua:1070-1094)
      sc->water_advance = (frame - tail->group_change) >= c.anim_frames;
      // 100 StochasticIntervalEpisodeEnding (component_library.lua:927-948):
      // _t was incremented by update() this step, so _t == step + 1.
      if (frame >= c.ee_min_frames && (step + 1) % c.ee_interval == 0) {
        if (philox_u53(wd.draw(RS_EPISODE_END, 0)) < c.thr_episode_end) tail->cont = 0;
      }
      // 4 AllNonselfCumulants.getCumulants (:535-545), 2 GlobalData.resetCumulants
      int total = 0;
      for (int p = 0; p < P; ++p) total += tail->flag0[p];
      for (int p = 0; p < P; ++p) sc->aux0[p] = (double)(total - tail->flag0[p]);
      for (int p = 0; p < P; ++p) { tail->flag0[p] = 0; tail->flag1[p] = 0; }

      // ---- flush 1: queued events in FIFO order (docs/advanced.md:43-52)
      // Avatar move (avatar_library.lua:155-203): turn, then moveRel
      for (int i = 0; i < P; ++i) {
        const int p = sc->order[0][i];
        const int turn = sc->act[p][A_TURN], move = sc->act[p][A_MOVE];
        if (turn != 0) tail->aori[p] = (uint8_t)((tail->aori[p] + turn + 4) & 3);
        if (move == 0 || !tail->aalive[p]) continue;
        const int dir = (tail->aori[p] + move - 1) & 3;
        int nx = tail->ax[p], ny = tail->ay[p];
        const int cur = ny * t.W + nx;
        bool ok = step_cell(t, nx, ny, dir, 1);
        const int ncell = ny * t.W + nx;
        if (ok && wd.at(t.avatar_layer, ncell) != 0) ok = false;
        if (!ok) { fire_enter(wd, p, cur); continue; }  // A3b: re-enters in place
        wd.at(t.avatar_layer, cur) = 0;
        wd.at(t.avatar_layer, ncell) = (uint8_t)t.alive_state[p];
        tail->ax[p] = (uint8_t)nx; tail->ay[p] = (uint8_t)ny;
        fire_enter(wd, p, ncell);
      }
      for (int i = 0; i < sc->n_fire[0]; ++i)
        beam(wd, sc->fire[0][i], HIT_ZAP, c.zap_length, c.zap_radius);
      for (int i = 0; i < sc->n_fire[1]; ++i)
        beam(wd, sc->fire[1][i], HIT_CLEAN, c.clean_length, c.clean_radius);
      // teleportToGroup(spawnGroup, aliveState), PICK_RANDOM orientation
      // (component_library.lua:336-354).  A5: uniform over the group's pieces
      // in creation order; an occupied target fails and is retried next frame.
      for (int i = 0; i < sc->n_respawn; ++i) {
        const int p = sc->respawn[i];
        const Philox4 d = wd.draw(RS_RESPAWN, (uint32_t)p);
        const int cell = t.spawn_cells[philox_bounded(d, (uint32_t)t.n_spawn)];
        if (wd.at(t.avatar_layer, cell) != 0) continue;
        tail->aalive[p] = 1;
        tail->ax[p] = (uint8_t)(cell % t.W); tail->ay[p] = (uint8_t)(cell / t.W);
        tail->achange[p] = frame;
        wd.at(t.avatar_layer, cell) = (uint8_t)t.alive_state[p];
        fire_enter(wd, p, cell);
        tail->aori[p] = (uint8_t)(d.x3 & 3u);
        tail->ctr[6]++;
      }
      if (sc->water_advance) tail->group_change = frame;
    }
    __syncthreads();
    // water Animation setStates (last events of flush 1), lane-parallel
    if (sc->water_advance) {
      for (int i = lane; i < c.n_water; i += 64) {
        const int cell = c.water_cells[i];
        const int s = wd.at(c.water_layer, cell);
        int k = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (s == c.s_water[q]) k = q;
        wd.at(c.water_layer, cell) = (uint8_t)c.s_water[(k + 1) & 3];
      }
    }
    __syncthreads();
    // ---- flush 2: setStates queued by the callbacks of flush 1
    if (lane == 0) {
      for (int i = 0; i < sc->n_pend_apple; ++i)  // apple -> appleWait (off-grid)
        wd.at(c.apple_layer, sc->pend_apple[i]) = 0;
      for (int p = 0; p < P; ++p) {               // zapped avatar -> playerWait
        if (!(sc->zapped_mask & (1u << p)) || !tail->aalive[p]) continue;
        wd.at(t.avatar_layer, tail->ay[p] * t.W + tail->ax[p]) = 0;
        tail->aalive[p] = 0;
        tail->achange[p] = frame;
      }
      for (int i = 0; i < sc->n_pend_dirt; ++i) { // dirt -> dirtWait
        const int cell = sc->pend_dirt[i];
        if (wd.at(c.dirt_layer, cell) != c.s_dirt) continue;
        if (wd.at(c.dirt_wait_layer, cell) != 0) continue;
        wd.at(c.dirt_layer, cell) = 0;
        wd.at(c.dirt_wait_layer, cell) = (uint8_t)c.s_dirt_wait;
        tail->aux_count--;  // DirtTracker:onStateChange
      }
      tail->frame = frame + 1;
      const int cont = tail->cont && step < t.max_frames;  // api_factory.lua:107-110
      tail->done = !cont;
    }
    __syncthreads();
    step_type = tail->done ? 2 : 1;
  }

  // ---- outputs: "N.REWARD", "N.READY_TO_SHOOT" (avatar_library.lua:737-744),
  // NUM_OTHERS_WHO_CLEANED_THIS_STEP (component_library.lua:786-803)
  if (lane < P) {
    const int p = lane;
    const double r = sc->reward[p];
    out.reward[(size_t)w * P + p] = r;
    double v = 1.0 - (double)tail->ztimer[p] / (double)c.zap_cooldown;
    out.ready[(size_t)w * P + p] = tail->aalive[p] ? (v > 0.0 ? v : 0.0) : 0.0;
    out.aux0[(size_t)w * P + p] = sc->aux0[p];
    out.position[((size_t)w * P + p) * 2 + 0] = tail->ax[p];
    out.position[((size_t)w * P + p) * 2 + 1] = tail->ay[p];
    out.orientation[(size_t)w * P + p] = tail->aori[p];
  }
  if (lane == 0) {
    double sum = 0.0;
    for (int p = 0; p < P; ++p) sum += sc->reward[p];
    out.collective[w] = sum;  // collective_reward_wrapper.py:49
    out.step_type[w] = step_type;
    out.discount[w] = step_type == 1 ? 1.0 : 0.0;
    tail->reward_fx += (uint32_t)(int32_t)(sum * 1024.0);
  }
  __syncthreads();
  for (int i = lane; i < nvec; i += 64)
    reinterpret_cast<uint4*>(gw)[i] = reinterpret_cast<const uint4*>(smem)[i];
}

}  // namespace

void launch_step_clean_up(const DevTables& t, const CleanUpTables& c,
                          uint8_t* state, int num_worlds, const int32_t* actions,
                          const uint8_t* reset_mask, int mode, int auto_reset,
                          const StepOutputs& out, hipStream_t stream) {
  const size_t lds = (size_t)t.world_stride + sizeof(Scratch);
  hipLaunchKernelGGL(k_step_clean_up, dim3(num_worlds), dim3(64), lds, stream, t,
                     c, state, actions, reset_mask, mode, auto_reset, out);
}
