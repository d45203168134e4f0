"""Pins the CPU oracle against everything the reference itself pins.

The reference ships no golden step/render vectors ("parity
unpinned").  What it does hold are small known-answer tests for the engine
primitives, written in Lua against dmlab2d:
  lua/modules/piece_movement_test.lua:69-89
  lua/modules/game_object_test.lua:182-188,267-411
They are restated here against the oracle's bare grid engine (no substrate
rules), plus the published Random123 known-answer vectors for Philox4x32-10.
"""
import numpy as np
import pytest

from meltingpot_amd import lower, pack
from oracle import oracle

N, E, S, W = 0, 1, 2, 3
KEEP_ORIGINAL = 1


def _bare_pack(width, objects, states, layers=3, groups=("testables", "spawnPoints", "inactives")):
  """objects: list of (x, y, state id); states: list of (layer, groupmask)."""
  hdr = np.zeros(lower.HDR_LEN, np.int32)
  hdr[lower.HDR_VERSION] = 1
  hdr[lower.HDR_SUBSTRATE] = 0
  hdr[lower.HDR_H], hdr[lower.HDR_W], hdr[lower.HDR_L] = 1, width, layers
  nst = len(states) + 1
  hdr[lower.HDR_NSTATES] = nst
  hdr[lower.HDR_NSPRITES] = 2
  hdr[lower.HDR_P] = 0
  hdr[lower.HDR_SPRITE] = 8
  hdr[lower.HDR_VL:lower.HDR_VB + 1] = (1, 1, 1, 1)
  hdr[lower.HDR_MAXFRAMES] = 1000
  hdr[lower.HDR_NOBJ] = len(objects)
  hdr[lower.HDR_NGROUPS] = len(groups)
  obj = np.zeros((len(objects), 4), np.int32)
  for i, (x, y, s) in enumerate(objects):
    obj[i] = (lower.KIND_STATIC, x, y, s)
  t = {
      "hdr": hdr,
      "state_layer": np.asarray([-1] + [s[0] for s in states], np.int32),
      "state_sprite": np.full(nst, -1, np.int32),
      "state_contact": np.full(nst, -1, np.int32),
      "state_groups": np.asarray([0] + [s[1] for s in states], np.uint32),
      "sprite_rgba": np.zeros((2, 4, 8, 8, 4), np.uint8),
      "sprite_flags": np.zeros(2, np.int32),
      "objects": obj,
      "avatar_alive_state": np.zeros(1, np.int32),
      "avatar_wait_state": np.zeros(1, np.int32),
      "view_sprite_map": np.zeros(2, np.int32),
      "hit_state": np.zeros(1, np.int32),
      "action_table": np.zeros(4, np.int32),
      "init_grid": np.zeros(layers * width, np.uint8),
      "init_spawn_cells": np.zeros(1, np.int32),
      "init_spawn_ptr": np.zeros(1, np.int32),
      "avatar_init_group": np.zeros(1, np.int32),
      "group_names": np.frombuffer(("\0".join(groups) + "\0").encode(), np.uint8).copy(),
  }
  return pack.dumps(t)


class Bare:
  """The oracle's bare engine driven the way the Lua KATs drive dmlab2d."""

  def __init__(self, blob):
    self.o = oracle.Oracle(blob, 1)
    self.L = oracle.lib()
    self.h = self.o.handle
    self.o.reset()

  def update(self):
    self.L.orc_grid_update(self.h)

  def pos(self, piece):
    return (self.L.orc_piece_x(self.h, piece), self.L.orc_piece_y(self.h, piece))

  def orient(self, piece):
    return self.L.orc_piece_orient(self.h, piece)

  def state(self, piece):
    return self.L.orc_piece_state(self.h, piece)


def _game_object_world():
  """game_object_test.lua:60-130: 5x1 grid, spawn point '.' at x=4, test object
  at (2,0) facing S in state1 (upperLayer; groups testables+spawnPoints)."""
  TESTABLES, SPAWN, INACTIVE = 1, 2, 4
  states = [(1, TESTABLES | SPAWN),      # 1: state1 on upperLayer
            (0, TESTABLES | INACTIVE),   # 2: state2 on lowerLayer
            (2, SPAWN)]                  # 3: spawnPoint on logic
  b = Bare(_bare_pack(5, [(4, 0, 3), (2, 0, 1)], states))
  obj = 1
  b.L.orc_q_set_orientation(b.h, obj, S)
  b.update()
  assert b.pos(obj) == (2, 0) and b.orient(obj) == S
  return b, obj, SPAWN


def test_piece_movement_move_abs():
  # piece_movement_test.lua:69-78: layout '   A ', moveAbs E: x 3 -> 4
  b = Bare(_bare_pack(5, [(3, 0, 1)], [(0, 0)]))
  assert b.pos(0) == (3, 0)
  b.L.orc_q_move_abs(b.h, 0, E)
  b.update()
  assert b.pos(0) == (4, 0)


def test_piece_movement_teleport():
  # piece_movement_test.lua:80-89
  b = Bare(_bare_pack(5, [(3, 0, 1)], [(0, 0)]))
  b.L.orc_q_teleport(b.h, 0, 1, 0)
  b.update()
  assert b.pos(0) == (1, 0)


def test_operations_are_queued_until_grid_update():
  # game_object_test.lua:182-188 (setUniqueStateWithoutGridUpdate)
  b, obj, _ = _game_object_world()
  b.L.orc_q_set_state(b.h, obj, 2)
  assert b.state(obj) == 1
  b.update()                       # :190-196 getUniqueStateAfterSet
  assert b.state(obj) == 2


def test_game_object_reset_restores_the_initial_state():
  # game_object_test.lua:380-411 (tests.reset): orientation, position and state
  # change under play; reset + start puts the object back at (2, 0), S, state1
  b, obj, _ = _game_object_world()
  assert b.orient(obj) == S and b.pos(obj) == (2, 0) and b.state(obj) == 1
  b.L.orc_q_set_orientation(b.h, obj, E)
  b.L.orc_q_move_abs(b.h, obj, E)
  b.L.orc_q_set_state(b.h, obj, 2)
  b.update()
  assert b.orient(obj) == E and b.pos(obj) == (3, 0) and b.state(obj) == 2
  b.o.reset()                                   # gameObject:reset() + gameObject:start(grid)
  b.L.orc_q_set_orientation(b.h, obj, S)        # (the test object's initial facing, makeTestGameObject)
  b.update()
  assert b.orient(obj) == S and b.pos(obj) == (2, 0) and b.state(obj) == 1


def test_game_object_move_abs():
  b, obj, _ = _game_object_world()  # game_object_test.lua:267-279
  b.L.orc_q_move_abs(b.h, obj, E)
  b.update()
  assert b.pos(obj) == (3, 0)


def test_game_object_move_rel_east_while_facing_south():
  b, obj, _ = _game_object_world()  # game_object_test.lua:281-293
  b.L.orc_q_move_rel(b.h, obj, E)
  b.update()
  assert b.pos(obj) == (1, 0)


def test_game_object_teleport():
  b, obj, _ = _game_object_world()  # game_object_test.lua:295-309
  b.L.orc_q_teleport(b.h, obj, 1, 0)
  b.L.orc_q_set_orientation(b.h, obj, W)
  b.update()
  assert b.pos(obj) == (1, 0) and b.orient(obj) == W


def test_game_object_teleport_to_group():
  b, obj, spawn = _game_object_world()  # game_object_test.lua:311-324
  b.L.orc_q_teleport_to_group(b.h, obj, spawn, 2, 2)
  b.update()
  assert b.pos(obj) == (4, 0) and b.state(obj) == 2


def test_game_object_teleport_to_group_keep_orientation():
  b, obj, spawn = _game_object_world()  # game_object_test.lua:326-345
  b.L.orc_q_teleport_to_group(b.h, obj, spawn, 2, KEEP_ORIGINAL)
  b.update()
  assert b.orient(obj) == S


def test_game_object_turn_3_is_counterclockwise():
  b, obj, _ = _game_object_world()  # game_object_test.lua:347-362: S -> E
  b.L.orc_q_turn(b.h, obj, 3)
  b.update()
  assert b.orient(obj) == E


def test_game_object_set_orientation():
  b, obj, _ = _game_object_world()  # game_object_test.lua:364-378
  b.L.orc_q_set_orientation(b.h, obj, E)
  b.update()
  assert b.orient(obj) == E


def test_game_object_combined_ops_in_one_update():
  b, obj, _ = _game_object_world()  # game_object_test.lua:380-411 (reset test)
  b.L.orc_q_set_orientation(b.h, obj, E)
  b.L.orc_q_move_abs(b.h, obj, E)
  b.L.orc_q_set_state(b.h, obj, 2)
  b.update()
  assert b.orient(obj) == E and b.pos(obj) == (3, 0) and b.state(obj) == 2


def test_move_into_occupied_cell_stays():
  # component_library.lua:292-309 "If there is a piece in the target location
  # at the time of the move then the piece stays where it is"
  b = Bare(_bare_pack(5, [(3, 0, 1), (4, 0, 1)], [(0, 0)]))
  b.L.orc_q_move_abs(b.h, 0, E)
  b.update()
  assert b.pos(0) == (3, 0) and b.pos(1) == (4, 0)


def test_bounded_topology_rejects_leaving_the_map():
  b = Bare(_bare_pack(5, [(4, 0, 1)], [(0, 0)]))
  b.L.orc_q_move_abs(b.h, 0, E)
  b.update()
  assert b.pos(0) == (4, 0)


@pytest.mark.parametrize("ctr,key,want", [
    ((0, 0, 0, 0), (0, 0),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2,
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
])
def test_philox4x32_10_random123_known_answers(ctr, key, want):
  # Random123 kat_vectors: "philox4x32 10 ..." (Salmon et al., SC'11)
  assert tuple(int(x) for x in oracle.philox(ctr, key)) == want
