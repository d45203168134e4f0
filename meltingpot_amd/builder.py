"""`builder(lab2d_settings, prefab_overrides, env_seed)` on the HIP engine.

The reference's cut point between Python and DMLab2D
(meltingpot/utils/substrates/builder.py:142-192) takes ANY settings dict — the
product of a config's `build(roles, config)` or one a caller has edited — plus
`prefab_overrides`, and returns a `dmlab2d.Environment`.  This module is that
entry point for the levels the engine implements: the settings are lowered at
run time (`lower.check_components` + `lower.lower`: no reference tree needed
once the settings dict is in hand, no "must be a committed map" restriction),
an engine is created on that pack, and the result is the `dmlab2d.Environment`
duck type of `lab2d_env` — what the reference's unmodified wrapper stack
(`build_substrate`) runs on.

    from meltingpot_amd import builder
    env = builder.builder(lab2d_settings,
                          prefab_overrides={"potential_apple": {"AppleGrow": {
                              "maxAppleGrowthRate": 0.1}}},
                          env_seed=7)

Nothing here computes a step: lowering is table building on the host, done once.
"""

from __future__ import annotations

import copy
from typing import Any, Mapping, Optional

import numpy as np

from meltingpot_amd import engine as engine_lib
from meltingpot_amd import lab2d_env, lower
from meltingpot_amd import pack as pack_lib
from meltingpot_amd import substrate as substrate_lib

Settings = Mapping[str, Any]

# What a level's avatars observe besides RGB, in the reference configs' order, and
# which of them travels as MP_OBS_AUX0 (configs/substrates/clean_up.py:813-832,
# commons_harvest__open.py:531-558, territory__rooms.py:84-104, coins.py:468-491,
# prisoners_dilemma_in_the_matrix__repeated.py:518-552).
_LEVEL_OBSERVATIONS = {
    "clean_up": (("RGB", "READY_TO_SHOOT", "NUM_OTHERS_WHO_CLEANED_THIS_STEP"),
                 "NUM_OTHERS_WHO_CLEANED_THIS_STEP"),
    "commons_harvest": (("RGB", "READY_TO_SHOOT"), None),
    "territory": (("RGB", "READY_TO_SHOOT"), None),
    "coins": (("RGB", "MISMATCHED_COIN_COLLECTED_BY_PARTNER"),
              "MISMATCHED_COIN_COLLECTED_BY_PARTNER"),
    "the_matrix": (("RGB", "INVENTORY", "READY_TO_SHOOT", "INTERACTION_INVENTORIES"), None),
    "coop_mining": (("RGB", "READY_TO_SHOOT"), None),
    "gift_refinements": (("RGB", "READY_TO_SHOOT", "INVENTORY"), None),
    "collaborative_cooking": (("RGB",), None),
    "externality_mushrooms": (("RGB", "READY_TO_SHOOT"), None),
}


def _plain(value):
  """A deep, writable copy of a settings tree as dicts / lists / scalars
  (ml_collections.ConfigDict and friends answer `to_dict` / `items`)."""
  if hasattr(value, "to_dict"):
    value = value.to_dict()
  if isinstance(value, Mapping):
    return {k: _plain(v) for k, v in value.items()}
  if isinstance(value, (list, tuple)):
    return [_plain(v) for v in value]
  return copy.deepcopy(value)


def _first_named_component(game_object, name: str):
  """game_object_utils.py:59-65."""
  for c in game_object["components"]:
    if c["component"] == name:
      return c
  raise ValueError(f"No component with name '{name}' found.")


def apply_prefab_overrides(lab2d_settings, prefab_overrides: Optional[Settings]) -> None:
  """builder.py:70-87: `{prefab: {component: {kwarg: value}}}` edits the kwargs of
  the first component of that name in `simulation.prefabs[prefab]`, in place."""
  sim = lab2d_settings["simulation"]
  sim.setdefault("gameObjects", [])
  if not prefab_overrides:
    return
  for prefab, override in prefab_overrides.items():
    for component, arg_overrides in override.items():
      for arg_name, arg_override in arg_overrides.items():
        if prefab not in sim["prefabs"]:
          raise ValueError(f"Prefab override for '{prefab}' given, but not "
                           "available in `prefabs`.")
        _first_named_component(sim["prefabs"][prefab], component)["kwargs"][arg_name] = (
            _plain(arg_override))


def maybe_build_and_add_avatar_objects(lab2d_settings) -> None:
  """builder.py:90-130 / game_object_utils.py:85-135: with an 'avatar' prefab and
  `buildAvatars` unset, one avatar object per player is derived from the prefab
  (sprite name and state sprite suffixed with the Lua index, its palette, its
  Avatar.index).  The default palettes are the reference's colour table
  (`colors.palette`), which this package does not carry: `playerPalettes` must be
  given in that case.  (None of the substrates lowered here takes this path: their
  configs put finished avatar objects into `simulation.gameObjects`.)"""
  sim = lab2d_settings["simulation"]
  build_here = "avatar" in sim["prefabs"]
  if sim.get("buildAvatars"):
    build_here = False
    if "avatar" not in sim["prefabs"]:
      raise ValueError("Deferring avatar building to Lua, yet no 'avatar' prefab given.")
    raise NotImplementedError("simulation.buildAvatars: avatars are not built on the Lua "
                              "side here; supply them in simulation.gameObjects")
  if not build_here:
    return
  num_players = int(lab2d_settings["numPlayers"])
  palettes = sim.get("playerPalettes")
  if not palettes:
    raise NotImplementedError(
        "an 'avatar' prefab without simulation.playerPalettes needs the reference's "
        "default colour table (utils/substrates/colors.py); pass playerPalettes")
  if len(palettes) < num_players:
    raise ValueError(f"Expected at least {num_players} player palettes, got {len(palettes)}.")
  for idx in range(num_players):
    obj = copy.deepcopy(sim["prefabs"]["avatar"])
    lua_index = idx + 1
    appearance = _first_named_component(obj, "Appearance")["kwargs"]
    sprite_name = appearance["spriteNames"][0]
    appearance["spriteNames"][0] = sprite_name + str(lua_index)
    for sc in _first_named_component(obj, "StateManager")["kwargs"]["stateConfigs"]:
      if sc.get("sprite") == sprite_name:
        sc["sprite"] = sprite_name + str(lua_index)
    appearance["palettes"][0] = palettes[idx]
    _first_named_component(obj, "Avatar")["kwargs"]["index"] = lua_index
    sim["gameObjects"].append(obj)


def config_of(level: str, tables) -> substrate_lib.SubstrateConfig:
  """The observation names and specs of a lowered pack (what `env_raw.
  observation_names()` and the specs of dmlab2d answer for the reference)."""
  if level not in _LEVEL_OBSERVATIONS:
    raise NotImplementedError(f"no engine for level {level!r}")
  hdr = tables["hdr"]
  P = int(hdr[lower.HDR_P])
  S = int(hdr[lower.HDR_SPRITE])
  H, W = int(hdr[lower.HDR_H]), int(hdr[lower.HDR_W])
  vh = int(hdr[lower.HDR_VF]) + int(hdr[lower.HDR_VB]) + 1
  vw = int(hdr[lower.HDR_VL]) + int(hdr[lower.HDR_VR]) + 1
  individual, aux0 = _LEVEL_OBSERVATIONS[level]
  A = substrate_lib.Array
  spec = {"RGB": A((vh * S, vw * S, 3), np.uint8, "RGB"),
          "WORLD.RGB": A((H * S, W * S, 3), np.uint8, "WORLD.RGB")}
  R = ((len(tables["mx_states"]) - 8) // 2 if "mx_states" in tables
       else int(tables["gr_i32"][7]) if "gr_i32" in tables else 0)
  for n in individual:
    if n == "INVENTORY":
      spec[n] = A((R,), np.float64, n)
    elif n == "INTERACTION_INVENTORIES":
      spec[n] = A((2, R), np.float64, n)
    elif n != "RGB":
      spec[n] = A((), np.float64, n)
  return substrate_lib.SubstrateConfig(
      name=level, action_set=({},), individual_observation_names=individual,
      global_observation_names=("WORLD.RGB",), timestep_spec=spec,
      valid_roles={"default"}, default_player_roles=("default",) * P, aux0_name=aux0)


def lower_settings(lab2d_settings: Settings, prefab_overrides: Optional[Settings] = None,
                   action_set=None):
  """The host half of `builder`: (level name, pack bytes, config) of a settings
  dict with the overrides applied — what the engine and the oracle are created on.
  `action_set`: the discrete actions mp_step looks up (a config's ACTION_SET); default:
  none but NOOP — `builder`'s environment takes raw action fields, as dmlab2d does."""
  assert "simulation" in lab2d_settings
  settings = _plain(lab2d_settings)          # "Copy config, so as not to modify it."
  apply_prefab_overrides(settings, prefab_overrides)
  maybe_build_and_add_avatar_objects(settings)
  # (locate_and_overwrite_level_directory, builder.py:132-139, points dmlab2d at the
  # Lua tree; the level is a compiled-in step function here: its name selects it)
  level = str(settings["levelName"]).rsplit("/", 1)[-1]
  settings["levelName"] = level
  tables = lower.lower(level, settings, list(action_set) if action_set else [{}])
  return level, pack_lib.dumps(tables), config_of(level, tables)


def builder(lab2d_settings: Settings, prefab_overrides: Optional[Settings] = None,
            env_seed: Optional[int] = None, *, device: int = 0,
            **settings) -> lab2d_env.Environment:
  """builder.py:142-192.  `env_seed`: as there, a random one when None; the
  reference's reset wrapper rebuilds the environment with env_seed + k for episode k,
  here episode k of the one world draws from a counter-based stream of the same seed
  (a reset never replays an episode either way).  The environment runs
  on a HIP engine on `device`, created on the pack the settings lower to."""
  del settings   # "Not currently used by DMLab2D."
  level, pack_bytes, config = lower_settings(lab2d_settings, prefab_overrides)
  players = len(config.default_player_roles)
  engine = engine_lib.Engine(
      pack_bytes, 1, device=device, auto_reset=True, num_players=players,
      base_seed=substrate_lib.resolve_env_seed(env_seed), literal_seed=True)
  return lab2d_env.Environment(level, config.default_player_roles, engine=engine, config=config)
