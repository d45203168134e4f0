"""ORACLE — test infrastructure only.

ctypes binding of oracle/liboracle.so (the scalar CPU restatement of the
reference's step+render path; provenance in oracle/engine.h).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product (`meltingpot_amd`) never does.  PARITY UNPINNED: see
DESIGN.md §oracle.
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
  """Compiles liboracle.so with gcc (Makefile in this directory)."""
  if force or not os.path.exists(_LIB_PATH):
    subprocess.run(["make", "-s", "-C", _HERE] + (["-B"] if force else []),
                   check=True)
  return _LIB_PATH


def lib():
  global _lib
  if _lib is None:
    build()
    L = ctypes.CDLL(_LIB_PATH)
    vp, u64, i32, u32 = (ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32,
                         ctypes.c_uint32)
    L.orc_create.restype = vp
    L.orc_create.argtypes = [vp, u64, u64]
    L.orc_create_players.restype = vp
    L.orc_create_players.argtypes = [vp, u64, u64, i32]
    L.orc_destroy.argtypes = [vp]
    L.orc_set_option.argtypes = [vp, i32, i32]
    L.orc_reset.argtypes = [vp]
    L.orc_step.restype = i32
    L.orc_step.argtypes = [vp, vp]
    L.orc_step_fields.restype = i32
    L.orc_step_fields.argtypes = [vp, vp]
    L.orc_place_avatar.restype = i32
    L.orc_place_avatar.argtypes = [vp, i32, i32, i32, i32, i32]
    L.orc_set_cell_state.restype = i32
    L.orc_set_cell_state.argtypes = [vp, i32, i32, i32, i32]
    L.orc_done.restype = i32
    L.orc_done.argtypes = [vp]
    L.orc_step_count.restype = i32
    L.orc_step_count.argtypes = [vp]
    for name in ("orc_rewards", "orc_ready_to_shoot", "orc_num_others_cleaned",
                 "orc_debug_metrics", "orc_zap_matrix"):
      getattr(L, name).argtypes = [vp, vp]
    L.orc_dump.argtypes = [vp, vp, vp, vp]
    L.orc_events.restype = i32
    L.orc_events.argtypes = [vp, vp, i32]
    L.orc_render_agent.argtypes = [vp, i32, vp]
    L.orc_render_world_rgb.argtypes = [vp, vp]
    L.orc_layer_view.argtypes = [vp, i32, vp]
    L.orc_inventories.restype = i32
    L.orc_inventories.argtypes = [vp, vp, vp]
    L.orc_matrix_cumulants.argtypes = [vp, vp]
    L.orc_interaction_rewards.argtypes = [vp, vp]
    L.orc_mt19937_64.restype = ctypes.c_uint64
    L.orc_mt19937_64.argtypes = [ctypes.c_uint64, i32]
    L.orc_mt19937_64_draw.restype = ctypes.c_uint64
    L.orc_mt19937_64_draw.argtypes = [ctypes.c_uint64, i32, i32, ctypes.c_uint64]
    for name in ("orc_piece_x", "orc_piece_y", "orc_piece_orient",
                 "orc_piece_state", "orc_avatar_piece"):
      getattr(L, name).restype = i32
      getattr(L, name).argtypes = [vp, i32]
    for name in ("orc_q_move_abs", "orc_q_move_rel", "orc_q_turn",
                 "orc_q_set_orientation", "orc_q_set_state"):
      getattr(L, name).argtypes = [vp, i32, i32]
    L.orc_q_teleport.argtypes = [vp, i32, i32, i32]
    L.orc_q_teleport_to_group.argtypes = [vp, i32, u32, i32, i32]
    L.orc_q_hit_beam.argtypes = [vp, i32, i32, i32, i32]
    L.orc_grid_update.argtypes = [vp]
    L.orc_philox.argtypes = [u32] * 6 + [vp]
    _lib = L
  return _lib


def philox(c, k):
  out = np.zeros(4, np.uint32)
  lib().orc_philox(*[int(x) for x in c], *[int(x) for x in k],
                   out.ctypes.data)
  return out


class Oracle:
  """One world of the CPU oracle."""

  def __init__(self, pack_bytes: bytes, world_seed: int, num_players: int = 0):
    """`num_players` = 0: as many as the pack holds; else the first
    `num_players` avatars of the pack play (num_players = len(roles))."""
    from meltingpot_amd import pack as pack_lib  # container format only
    self._L = lib()
    self._buf = ctypes.create_string_buffer(pack_bytes, len(pack_bytes))
    self._h = self._L.orc_create_players(
        self._buf, len(pack_bytes), ctypes.c_uint64(world_seed & (2**64 - 1)),
        int(num_players))
    if not self._h:
      raise ValueError("oracle: bad pack or unsupported substrate")
    t = pack_lib.loads(pack_bytes)
    hdr = t["hdr"]
    self.H, self.W, self.L, self.P = (int(hdr[2]), int(hdr[3]), int(hdr[4]),
                                      int(hdr[7]))
    if num_players:
      if not 0 < num_players <= self.P:
        raise ValueError(f"oracle: {num_players} players, the pack holds {self.P}")
      self.P = int(num_players)
    elif 0 < int(hdr[20]) <= self.P:   # MPK_HDR_DEFAULT_P
      self.P = int(hdr[20])
    self.view = (int(hdr[10]) + int(hdr[11]) + 1, int(hdr[12]) + int(hdr[13]) + 1)
    self.tables = t

  def close(self):
    if self._h:
      self._L.orc_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  @property
  def handle(self):
    return self._h

  # engine assumption switches: name -> (index, default)
  OPTIONS = {
      "A3b_blocked_move_reenters": (0, 1),
      "A4_beam_marks_blocked": (1, 1),
      "A6_dead_view_black": (2, 1),
      "A1_shuffle_order": (3, 1),
      "A2_flush_count": (4, 128),
      "A5_teleport_free_only": (5, 0),
      # A10s: one serial mt19937_64 per world consumed in call order (oracle/mt19937_64.h)
      # instead of the counter-based generator (A10) — set BEFORE reset(); and its
      # conversions: uniform_int_distribution's method (0 Lemire, 1 scaling + rejection),
      # Fisher-Yates from the back
      "A10s_serial_mt19937": (6, 0),
      "A10s_int_method": (7, 0),
      "A10s_shuffle_back": (8, 0),
  }

  def set_option(self, which, value: int):
    """`which`: an index or a name of `OPTIONS`."""
    if isinstance(which, str):
      which = self.OPTIONS[which][0]
    self._L.orc_set_option(self._h, which, value)

  def place_avatar(self, p: int, x: int, y: int, orient: int, alive: bool = True) -> bool:
    """Puts avatar p where a recorded trajectory has it (trace fitting)."""
    return bool(self._L.orc_place_avatar(self._h, p, int(x), int(y), int(orient), int(alive)))

  def set_cell_state(self, layer: int, x: int, y: int, state: int) -> bool:
    """(test hook) the piece at (layer, x, y) in `state` at once (oracle_api.c)."""
    return bool(self._L.orc_set_cell_state(self._h, int(layer), int(x), int(y), int(state)))

  def reset(self):
    self._L.orc_reset(self._h)

  def step(self, actions) -> bool:
    a = np.ascontiguousarray(actions, np.int32)
    assert a.shape == (self.P,)
    return bool(self._L.orc_step(self._h, a.ctypes.data))

  def step_fields(self, fields) -> bool:
    """One step from raw action fields [P, A] in actionOrder (orc_step_fields)."""
    a = np.ascontiguousarray(fields, np.int32)
    assert a.shape == (self.P, int(self.tables["hdr"][21]))   # MPK_HDR_NFIELDS
    return bool(self._L.orc_step_fields(self._h, a.ctypes.data))

  @property
  def done(self) -> bool:
    return bool(self._L.orc_done(self._h))

  def _vec(self, fn):
    out = np.zeros(max(self.P, 1), np.float64)
    fn(self._h, out.ctypes.data)
    return out[:self.P]

  def rewards(self):
    return self._vec(self._L.orc_rewards)

  def ready_to_shoot(self):
    return self._vec(self._L.orc_ready_to_shoot)

  def num_others_cleaned(self):
    return self._vec(self._L.orc_num_others_cleaned)

  def debug_metrics(self):
    """[4, P]: PLAYER_CLEANED, PLAYER_ATE_APPLE, NUM_OTHERS_PLAYER_ZAPPED_THIS_STEP,
    NUM_OTHERS_WHO_ATE_THIS_STEP (clean_up.py:751-784)."""
    out = np.zeros((4, max(self.P, 1)), np.float64)
    buf = np.zeros(4 * self.P, np.float64)
    self._L.orc_debug_metrics(self._h, buf.ctypes.data)
    return buf.reshape(4, self.P)

  def zap_matrix(self):
    """[P, P] playerZapMatrix(zapped, zapper) of the last step."""
    buf = np.zeros(self.P * self.P, np.float64)
    self._L.orc_zap_matrix(self._h, buf.ctypes.data)
    return buf.reshape(self.P, self.P)

  def inventories(self):
    """*_in_the_matrix: ("N.INVENTORY" [P, R], "N.INTERACTION_INVENTORIES" [P, 2, R])."""
    inv = np.zeros((self.P, 3), np.float64)
    inter = np.zeros((self.P, 2, 3), np.float64)
    a, b = np.zeros(self.P * 3), np.zeros(self.P * 6)
    R = self._L.orc_inventories(self._h, a.ctypes.data, b.ctypes.data)
    assert R > 0, "not an *_in_the_matrix pack"
    return a[:self.P * R].reshape(self.P, R), b[:self.P * 2 * R].reshape(self.P, 2, R)

  def matrix_cumulants(self):
    """[P, 1 + 3 R]: INTERACTED_THIS_STEP, then per class COLLECTED_RESOURCE_k,
    DESTROYED_RESOURCE_k, ARGMAX_INTERACTION_INVENTORY_WAS_k (the_matrix.py:22-66)."""
    R = self.inventories()[0].shape[1]
    out = np.zeros(self.P * (1 + 3 * R), np.float64)
    self._L.orc_matrix_cumulants(self._h, out.ctypes.data)
    return out.reshape(self.P, 1 + 3 * R)

  def interaction_rewards(self):
    """[P, 2]: (row_reward, col_reward) of the 'interaction' event each player last
    took part in (the_matrix/components.lua:785-797)."""
    out = np.zeros(self.P * 2, np.float64)
    self._L.orc_interaction_rewards(self._h, out.ctypes.data)
    return out.reshape(self.P, 2)

  def events(self):
    """api:events of the last reset / step: sorted list of (type, a, b)."""
    buf = np.zeros((256, 3), np.int32)
    n = self._L.orc_events(self._h, buf.ctypes.data, 256)
    assert n <= 256, "oracle event log overflow"
    return sorted(tuple(int(v) for v in row) for row in buf[:n])

  def dump(self):
    grid = np.zeros((self.L, self.H, self.W), np.uint8)
    avat = np.zeros((max(self.P, 1), 8), np.int32)
    glob = np.zeros(8, np.int32)
    self._L.orc_dump(self._h, grid.ctypes.data, avat.ctypes.data,
                     glob.ctypes.data)
    return grid, avat[:self.P], glob

  def render_agent(self, p: int):
    vw, vh = self.view
    out = np.zeros((vh * 8, vw * 8, 3), np.uint8)
    self._L.orc_render_agent(self._h, p, out.ctypes.data)
    return out

  def layer_view(self, p: int):
    """"N.LAYER": int32 [vh, vw, L] (A17)."""
    vw, vh = self.view
    out = np.zeros((vh, vw, self.L), np.int32)
    self._L.orc_layer_view(self._h, p, out.ctypes.data)
    return out

  def render_world(self):
    out = np.zeros((self.H * 8, self.W * 8, 3), np.uint8)
    self._L.orc_render_world_rgb(self._h, out.ctypes.data)
    return out
