"""The reference's own Lua known-answer tests for the LOWERING and the SCHEDULER,
restated 1:1 against this repo's restatements of the code they test.

(tests/test_oracle_reference_kats.py does the same for the grid-engine
primitives.)  The Lua tests cannot run here — they need dmlab2d's Lua runtime —
so each test below cites the Lua test it restates; names, inputs and expected
values are the reference's.

  lua/modules/prefab_utils_test.lua:67-193     -> lower.build_game_object_configs
  lua/modules/updater_registry_test.lua:79-245 -> schedule.UpdaterRegistry
  lua/modules/avatar_library_test.lua:72-101   -> lower.lower_common (hits, renderOrder)
  lua/modules/component_library_test.lua:73-94 -> lower._Sprites (custom sprites)

and the frame schedule of the four lowered levels — derived by
`schedule.level_update_order` from the reference configs' own component lists —
against the order in which the oracle actually runs its updaters.
"""
import ctypes
import os

import numpy as np
import pytest

import util
from meltingpot_amd import lower, refshim
from oracle import schedule

HAVE_REFERENCE = os.path.isdir(refshim.DEFAULT_REFERENCE_ROOT)


# ------------------------------------------------------------ prefab_utils_test.lua
# `when(random).choice(rngState, anyValue).thenCall(function(x, y) return y[2] end)`
# (:30): the mocked random:choice returns the SECOND element of the list.
_CHOICE = lambda lst: lst[1]


def _prefabs(*names):
  return {n: {"name": n} for n in names}


def test_build_game_object_from_named_prefab():           # :67-75
  got = lower.build_game_object_configs("a", _prefabs("prefab"), {"a": "prefab"})
  assert got == [("prefab", 0, 0)]


def test_build_two_rows_game_objects_from_named_prefab():  # :77-86
  got = lower.build_game_object_configs("a\na", _prefabs("prefab"), {"a": "prefab"})
  assert got == [("prefab", 0, 0), ("prefab", 0, 1)]        # position = {col, row}


def test_build_two_cols_game_objects_from_named_prefab():  # :88-97
  got = lower.build_game_object_configs("aa", _prefabs("prefab"), {"a": "prefab"})
  assert got == [("prefab", 0, 0), ("prefab", 1, 0)]


def test_build_two_different_game_objects_from_named_prefab():   # :99-113
  got = lower.build_game_object_configs("ba", _prefabs("prefabA", "prefabB"),
                                        {"a": "prefabA", "b": "prefabB"})
  assert got == [("prefabB", 0, 0), ("prefabA", 1, 0)]


def test_build_game_object_not_found_ignored():            # :115-123
  assert lower.build_game_object_configs("b", _prefabs("prefab"), {"a": "prefab"}) == []


def test_build_game_object_from_all_spec():                # :125-140
  cpm = {"x": {"type": "all", "list": ["prefabB", "prefabA"]}}
  got = lower.build_game_object_configs("x", _prefabs("prefabA", "prefabB"), cpm)
  assert got == [("prefabB", 0, 0), ("prefabA", 0, 0)]


def test_build_game_object_from_choice_spec():             # :142-155
  cpm = {"x": {"type": "choice", "list": ["prefabB", "prefabA"]}}
  got = lower.build_game_object_configs("x", _prefabs("prefabA", "prefabB"), cpm, _CHOICE)
  assert got == [("prefabA", 0, 0)]


def test_build_game_object_from_nested_spec_choice_all():  # :157-174
  cpm = {"x": {"type": "choice",
               "list": ["prefabB", {"type": "all", "list": ["prefabA", "prefabB"]}]}}
  got = lower.build_game_object_configs("x", _prefabs("prefabA", "prefabB"), cpm, _CHOICE)
  assert got == [("prefabA", 0, 0), ("prefabB", 0, 0)]


def test_build_game_object_from_nested_spec_all_choice():  # :176-193
  cpm = {"x": {"type": "all",
               "list": ["prefabA", {"type": "choice", "list": ["prefabB", "prefabC"]}]}}
  got = lower.build_game_object_configs("x", _prefabs("prefabA", "prefabB", "prefabC"), cpm,
                                        _CHOICE)
  assert got == [("prefabA", 0, 0), ("prefabC", 0, 0)]


def test_leading_newlines_are_stripped_and_unknown_prefab_names_assert():
  # _visitText (prefab_utils.lua:113-131) skips leading newlines only; a name that
  # is not in `prefabs` is an assertion (:68-69), as in the reference
  got = lower.build_game_object_configs("\n\na\n a", _prefabs("p"), {"a": "p"})
  assert got == [("p", 0, 0), ("p", 1, 1)]
  with pytest.raises(AssertionError):
    lower.build_game_object_configs("a", _prefabs("p"), {"a": "q"})


def test_the_lowering_enumerates_a_choice_characters_outcomes_in_list_order():
  """What the packs carry for a per-episode 'choice' character: one alternative
  per list entry, each expanded as the reference would expand the chosen entry."""
  prefabs = _prefabs("A", "B", "C")
  spec = {"type": "choice", "list": ["B", {"type": "all", "list": ["A", "C"]}, "B"]}
  assert lower._alternatives(spec, prefabs) == [["B"], ["A", "C"], ["B"]]
  for k in range(3):   # outcome k == random:choice returning list[k]
    assert lower.build_game_object_configs("x", prefabs, {"x": spec}, lambda l, k=k: l[k]) == [
        (n, 0, 0) for n in lower._alternatives(spec, prefabs)[k]]


# --------------------------------------------------------- updater_registry_test.lua
def _test_game_object(object_id):   # makeTestGameObject (:46-74)
  return schedule.GameObjectStates(object_id, {"state1": ["spawnPoints"], "state2": []})


def test_register_single_simple_updater():                 # :79-85
  r = schedule.UpdaterRegistry()
  r.register_updater(lambda: None)
  assert r.get_sorted_priorities() == [100]


def test_register_single_updater_with_priority():          # :87-94
  r = schedule.UpdaterRegistry()
  r.register_updater(lambda: None, priority=53)
  assert r.get_sorted_priorities() == [53]


def test_register_multiple_updaters_with_same_priority():  # :96-111
  r = schedule.UpdaterRegistry()
  for _ in range(3):
    r.register_updater(lambda: None, priority=53)
  assert r.get_sorted_priorities() == [53]


def test_register_multiple_updaters_with_different_priority():   # :113-131
  r = schedule.UpdaterRegistry()
  r.register_updater(lambda: None)
  r.register_updater(lambda: None, priority=42)
  r.register_updater(lambda: None, priority=123)
  r.register_updater(lambda: None, priority=42)
  assert r.get_sorted_priorities() == [123, 100, 42]


def test_uniquify_ids_default():                           # :133-144
  go = _test_game_object("OID_1")
  r = schedule.UpdaterRegistry()
  r.register_updater(lambda: None)
  r.uniquify_states_and_add_groups(go)
  assert go.get_groups_for_state("state1") == ["spawnPoints", "UPDATER_GRP_____Updater#0"]
  assert go.get_groups_for_state("state2") == ["UPDATER_GRP_____Updater#0"]


def test_uniquify_ids_with_group_prefix():                 # :146-159
  go = _test_game_object("OID_1")
  r = schedule.UpdaterRegistry()
  r.set_group_prefix("my_prefix")
  r.register_updater(lambda: None)
  r.uniquify_states_and_add_groups(go)
  assert go.get_groups_for_state("state1") == ["spawnPoints", "UPDATER_GRP__my_prefix_Updater#0"]
  assert go.get_groups_for_state("state2") == ["UPDATER_GRP__my_prefix_Updater#0"]


def test_uniquify_ids_with_multiple_group_prefixes():      # :161-180
  go = _test_game_object("OID_1")
  r = schedule.UpdaterRegistry()
  r.set_group_prefix("my_prefix")
  r.register_updater(lambda: None)
  r.set_group_prefix("another")   # "pretend we are in another component"
  r.register_updater(lambda: None)
  r.uniquify_states_and_add_groups(go)
  assert go.get_groups_for_state("state1") == [
      "spawnPoints", "UPDATER_GRP__my_prefix_Updater#0", "UPDATER_GRP__another_Updater#0"]
  assert go.get_groups_for_state("state2") == [
      "UPDATER_GRP__my_prefix_Updater#0", "UPDATER_GRP__another_Updater#0"]


def test_uniquify_ids_with_multiple_group_prefixes_none_new():   # :182-196
  go = _test_game_object("OID_1")
  r = schedule.UpdaterRegistry()
  r.register_updater(lambda: None, group="spawnPoints")
  r.set_group_prefix("my_prefix")
  r.register_updater(lambda: None, group="preexisting")
  r.uniquify_states_and_add_groups(go)
  assert go.get_groups_for_state("state1") == ["spawnPoints"]
  assert go.get_groups_for_state("state2") == []


def test_uniquify_ids_with_multiple_group_prefixes_two_game_objects():   # :198-245
  def set_updates(go):
    r = schedule.UpdaterRegistry()
    r.set_group_prefix("my_prefix")
    r.register_updater(lambda: None, priority=90)
    r.set_group_prefix("another")
    r.register_updater(lambda: None)
    r.uniquify_states_and_add_groups(go)
    return r
  go1, go2 = _test_game_object("OID_1"), _test_game_object("OID_2")
  r1, r2 = set_updates(go1), set_updates(go2)
  for go in (go1, go2):
    assert go.get_groups_for_state("state1") == [
        "spawnPoints", "UPDATER_GRP__my_prefix_Updater#0", "UPDATER_GRP__another_Updater#0"]
    assert go.get_groups_for_state("state2") == [
        "UPDATER_GRP__my_prefix_Updater#0", "UPDATER_GRP__another_Updater#0"]
  merged = schedule.UpdaterRegistry()
  merged.merge_with(r1)
  merged.merge_with(r2)
  assert merged.get_sorted_priorities() == [100, 90]
  order = []
  merged.add_update_order(order)
  assert order == ["_priority_100_UPDATER_GRP__another_Updater#0",
                   "_priority_90_UPDATER_GRP__my_prefix_Updater#0"]


def test_uniquify_ids_some_state():                        # :247-261
  go = _test_game_object("OID_1")
  r = schedule.UpdaterRegistry()
  r.register_updater(lambda: None, state="state1")
  r.uniquify_states_and_add_groups(go)
  assert go.get_groups_for_state("state1") == ["spawnPoints", "UPDATER_GRP_____Updater#0"]
  assert go.get_groups_for_state("state2") == []


# ------------------------------------- utils/substrates/game_object_utils_test.py (ParseMapTest)
# The reference's PYTHON reading of an ASCII map (get_game_object_positions_from_map,
# game_object_utils.py: the row index starts after the leading newline, x = column) against
# the lowering's map visitor — the same inputs and expected positions as the reference test.
def _positions(ascii_map, char):
  return sorted((x, y) for x, y, _ in lower._visit_map(ascii_map, {char: char}))


@pytest.mark.parametrize("ascii_map,char,exp_len", [
    ("\nHello", "H", 1), ("\nHello", "h", 0), ("\nHello", "l", 2),
    ("\nHello\nWorld", "l", 3), ("\nHello\nWorld", "o", 2), ("\nHello\nWorld", "d", 1),
    ("\nHello\nWorld", "W", 1), ("\nWWWW\nW AW\nWWWW", "A", 1),
    ("\nWWWW\nW AW\nWWWW", "W", 10), ("\nWWWW\nW AW\nWWWW", "P", 0)])
def test_get_positions_length(ascii_map, char, exp_len):
  """game_object_utils_test.py:28-44 (ParseMapTest.test_get_positions_length)."""
  assert len(_positions(ascii_map, char)) == exp_len


def test_get_positions():
  """game_object_utils_test.py:46-92 (ParseMapTest.test_get_positions): 'A' at (2, 1), the
  blanks at (1, 1), (3, 1), (4, 1), the fourteen walls around them."""
  ascii_map = "\nWWWWWW\nW A  W\nWWWWWW\n"
  assert _positions(ascii_map, "A") == [(2, 1)]
  assert _positions(ascii_map, " ") == [(1, 1), (3, 1), (4, 1)]
  walls = ([(x, 0) for x in range(6)] + [(0, 1), (5, 1)] + [(x, 2) for x in range(6)])
  assert _positions(ascii_map, "W") == sorted(walls)


@pytest.mark.skipif(not HAVE_REFERENCE, reason="needs the reference tree")
@pytest.mark.parametrize("name,players", [("clean_up", 7), ("territory__rooms", 9),
                                          ("commons_harvest__open", 16)])
def test_map_visitor_agrees_with_the_reference_function(name, players):
  """... and the reference's own function (imported from the reference tree) on the ASCII
  maps of the lowered configs: every character at the same (x, y)."""
  import sys
  settings, _, _ = refshim.build_settings(name, ("default",) * players)
  gou = sys.modules["meltingpot.utils.substrates.game_object_utils"]
  ascii_map = settings["simulation"]["map"]
  for ch in sorted(set(ascii_map) - {"\n"}):
    ref = sorted((t.position.x, t.position.y)
                 for t in gou.get_game_object_positions_from_map(ascii_map, ch))
    assert ref == _positions(ascii_map, ch), (name, ch)


# ------------------------------------------- avatar_library_test / component_library_test
def _tiny_level(avatar_components, scene_components=()):
  """A 2 x 1 level: a wall and a spawn point, one avatar."""
  def states(*cfgs, initial):
    return {"component": "StateManager",
            "kwargs": {"initialState": initial, "stateConfigs": list(cfgs)}}
  transform = {"component": "Transform", "kwargs": {"position": (0, 0), "orientation": "N"}}
  wall = {"name": "wall", "components": [
      states({"state": "wall", "layer": "upperPhysical", "sprite": "Wall"}, initial="wall"),
      transform,
      {"component": "Appearance", "kwargs": {"renderMode": "colored_square",
                                            "spriteNames": ["Wall"],
                                            "spriteRGBColors": [(9, 9, 9)]}}]}
  spawn = {"name": "spawn", "components": [
      states({"state": "spawnPoint", "layer": "logic", "groups": ["spawnPoints"]},
             initial="spawnPoint"), transform]}
  avatar = {"name": "avatar", "components": [
      states({"state": "player", "layer": "upperPhysical", "sprite": "Avatar",
              "contact": "avatar"}, {"state": "playerWait"}, initial="player"),
      transform,
      {"component": "Appearance", "kwargs": {"renderMode": "colored_square",
                                            "spriteNames": ["Avatar"],
                                            "spriteRGBColors": [(1, 2, 3)]}},
      {"component": "Avatar", "kwargs": {
          "index": 1, "aliveState": "player", "waitState": "playerWait",
          "spawnGroup": "spawnPoints", "actionOrder": ["move", "turn", "fireZap"],
          "actionSpec": {"move": {"default": 0, "min": 0, "max": 4},
                         "turn": {"default": 0, "min": -1, "max": 1},
                         "fireZap": {"default": 0, "min": 0, "max": 1}}, "view": {"left": 5, "right": 5, "forward": 9, "backward": 1,
                                     "centered": False}}}] + list(avatar_components)}
  scene = {"name": "scene", "components": [
      states({"state": "scene"}, initial="scene"), transform] + list(scene_components)}
  return {"levelName": "kat", "numPlayers": 1, "spriteSize": 8, "topology": "BOUNDED",
          "maxEpisodeLengthFrames": 10,
          "simulation": {"map": "WP", "gameObjects": [avatar], "scene": scene,
                         "prefabs": {"wall": wall, "spawn": spawn},
                         "charPrefabMap": {"W": "wall", "P": "spawn"}}}


def test_zapper_adds_its_hit_and_appends_its_layer_to_the_render_order():
  """avatar_library_test.lua:72-101 (tests.zapper): Zapper:addHits gives
  hits = {zapHit = {layer = 'beamZap', sprite = 'BeamZap'}} and appends 'beamZap'
  to the renderOrder — here on top of the BaseSimulation order
  (base_simulation.lua:263-271)."""
  zapper = {"component": "Zapper", "kwargs": {
      "cooldownTime": 86, "beamLength": 13, "beamRadius": 9, "framesTillRespawn": 92,
      "penaltyForBeingZapped": -5, "rewardForZapping": 5}}
  t = lower.lower_common(_tiny_level([zapper]))
  assert t["_hits"] == [("zapHit", "beamZap", "BeamZap")]
  assert t["_layers"] == list(lower.BASE_RENDER_ORDER) + ["beamZap"]
  without = lower.lower_common(_tiny_level([]))
  assert without["_hits"] == [] and without["_layers"] == list(lower.BASE_RENDER_ORDER)
  # the BeamZap sprite is registered with the default colour (avatar_library.lua:579,606)
  i = t["_sprites"].index("BeamZap")
  assert tuple(t["sprite_rgba"][i, 0, 0, 0]) == (252, 252, 106, 255)


def test_additional_sprites_registers_its_custom_sprite_names():
  """component_library_test.lua:73-94 (tests.additionalSprites): the component's
  customSpriteNames become sprites of the tile set / custom sprites of the world."""
  shape = "\n".join(["@@@@@@@@"] + ["@xxAAxx@"] * 6 + ["@@@@@@@@"])
  palette = {"A": (20, 40, 60, 255), "B": (5, 5, 5, 255), "x": (0, 0, 0, 0),
             "@": (23, 11, 19, 255)}
  extra = {"component": "AdditionalSprites", "kwargs": {
      "renderMode": "ascii_shape", "customSpriteNames": ["Sprite1", "Sprite2"],
      "customSpriteShapes": [shape, shape], "customPalettes": [palette, palette],
      "customNoRotates": [True, True]}}
  t = lower.lower_common(_tiny_level([extra]))
  assert "Sprite1" in t["_sprites"] and "Sprite2" in t["_sprites"]
  i = t["_sprites"].index("Sprite1")
  assert tuple(t["sprite_rgba"][i, 0, 0, 0]) == palette["@"]      # top-left pixel, facing N
  assert tuple(t["sprite_rgba"][i, 2, 1, 3]) == palette["A"]      # noRotate: same art facing S


# ------------------------------------------------- the levels' schedules vs the oracle
def _oracle_trace(o):
  from oracle import oracle as oracle_lib
  L = oracle_lib.lib()
  L.orc_updater_trace.restype = ctypes.c_int
  L.orc_updater_trace.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
  buf = ctypes.create_string_buffer(8192)
  n = L.orc_updater_trace(o.handle, buf, 8192)
  out = []
  for line in buf.raw[:n].decode().splitlines():
    prio, tag = line.split(":", 1)
    out.append((int(prio), tag))
  return out


def _collapse(order):
  """Folds the per-state / per-count registrations of one component into one
  entry, as the oracle runs them (Animation: one updater per water frame state,
  DensityRegrow: one per neighbour count; all share a priority)."""
  out = []
  for prio, tag in order:
    head = tag.split(".")[0]
    if head in ("Animation", "DensityRegrow", "FixedRateRegrow"):
      tag = {"Animation": "Animation", "DensityRegrow": "DensityRegrow.sprout",
             "FixedRateRegrow": "FixedRateRegrow.regrow"}[head]
    if (prio, tag) not in out:
      out.append((prio, tag))
  return out


@pytest.mark.skipif(not HAVE_REFERENCE, reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name,players", [
    ("clean_up", 7), ("commons_harvest__open", 7), ("territory__rooms", 9), ("coins", 2),
    ("coop_mining", 6), ("gift_refinements", 6),
    ("collaborative_cooking__cramped", 2), ("collaborative_cooking__figure_eight", 6),
    ("externality_mushrooms__dense", 5),
    ("prisoners_dilemma_in_the_matrix__repeated", 2),
    ("running_with_scissors_in_the_matrix__arena", 8),
    ("running_with_scissors_in_the_matrix__one_shot", 2)])
def test_the_oracle_runs_its_updaters_in_the_order_the_registry_gives(name, players):
  """The reference's configs list the components; `schedule.COMPONENT_UPDATERS`
  holds what each registers (priority, state, startFrame, probability, cited per
  Lua line); UpdaterRegistry semantics order them (priority descending,
  updater_registry.lua:166-173,260-273).  The oracle must run exactly that
  sequence every frame."""
  import random
  from meltingpot_amd import engine
  from oracle import oracle as oracle_lib
  random.seed(0)
  settings, _, _ = refshim.build_settings(name, ("default",) * players)
  sim = settings["simulation"]
  objects = ([sim["scene"]] if "scene" in sim else []) + list(sim["gameObjects"])
  objects += [sim["prefabs"][p] for p, _, _ in lower.build_game_object_configs(
      sim["map"], sim["prefabs"], sim["charPrefabMap"], choice=lambda l: l[0])]
  want = _collapse(schedule.level_update_order(objects, settings["levelName"]))
  assert [p for p, _ in want] == sorted((p for p, _ in want), reverse=True)
  o = oracle_lib.Oracle(engine.load_pack(name), util.world_seed(0), players)
  o.reset()
  o.step(np.zeros(players, np.int32))
  got = _oracle_trace(o)
  # same multiset per priority, priorities descending; inside one priority the
  # reference's order is unspecified (pairs()): A11 fixes it to
  # registration order, which is what both sides must show
  assert got == want, (got, want)


@pytest.mark.skipif(not HAVE_REFERENCE, reason="reference tree not present (GPU box)")
def test_component_updater_tables_carry_the_configs_constants():
  """startFrame / probability of the registrations come from the component kwargs
  the pack's thresholds were lowered from."""
  settings, _, _ = refshim.build_settings("clean_up", ("default",) * 7)
  avatar = settings["simulation"]["gameObjects"][0]
  kw = {c["component"]: c.get("kwargs", {}) for c in avatar["components"]}
  regs = dict(schedule.COMPONENT_UPDATERS["Zapper"](kw["Zapper"]))
  assert regs["Zapper.zap"] == {"priority": 140}
  assert regs["Zapper.respawn"]["priority"] == 135
  assert regs["Zapper.respawn"]["start_frame"] == 50            # clean_up.py:707-716
  scene = {c["component"]: c.get("kwargs", {}) for c in settings["simulation"]["scene"]["components"]}
  ee = dict(schedule.COMPONENT_UPDATERS["StochasticIntervalEpisodeEnding"](
      scene["StochasticIntervalEpisodeEnding"]))
  assert ee["StochasticIntervalEpisodeEnding.maybeEndEpisode"] == {"start_frame": 1000}
