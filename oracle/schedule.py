"""TEST INFRASTRUCTURE (it lives next to the oracle for that reason; nothing in
meltingpot_amd/ imports it): the per-frame updater schedule of a level, restated
from the reference, to CHECK the order the kernels and the oracle hard-code.

In the reference every component registers its engine-driven callbacks with an
`UpdaterRegistry` (lua/modules/updater_registry.lua:114-303): a priority (100
if not given), an optional state / group the updater is restricted to, a
`startFrame` (frames the piece must have spent in its state) and a
probability.  `grid:update` then runs the updaters in priority-DESCENDING order
(`getSortedPriorities`, :166-173; `addUpdateOrder`, :260-273).  The HIP step
kernels (meltingpot_amd/csrc/step_*.h) and the CPU oracle (oracle/*.c) have that
order written into their code by hand; this module derives it independently from
the reference's registration calls, and the tests compare:

  * `UpdaterRegistry` restates the registry itself, call for call, so that the
    reference's own known-answer tests (updater_registry_test.lua:87-245) run
    against it (tests/test_reference_kats.py);
  * `COMPONENT_UPDATERS` lists, per Lua component, the `registerUpdater` calls
    of its `registerUpdaters` method, in source order, each citing its line;
  * `level_update_order(settings)` registers the components of a level's
    objects in creation order (scene, avatars, map objects; components in
    config order — A11: the reference iterates them with `pairs()`, so any fixed
    order conforms) and returns the frame's schedule.  The oracle logs the
    updaters it runs (`orc_updater_trace`); tests compare the two.

Nothing here runs on the hot path.
"""

from __future__ import annotations

from typing import Any, Callable, Dict, List, Mapping, Optional, Sequence, Tuple


class GameObjectStates:
  """What `UpdaterRegistry:uniquifyStatesAndAddGroups` needs of a game object:
  its states, their groups and its id (game_object.lua: getAllStates,
  getGroupsForState, getUniqueState)."""

  def __init__(self, object_id: str, state_groups: Mapping[str, Sequence[str]]):
    self.id = object_id
    self._groups = {s: list(g) for s, g in state_groups.items()}

  def get_all_states(self) -> List[str]:
    return list(self._groups)

  def get_groups_for_state(self, state: str) -> List[str]:
    return self._groups[state]

  def get_unique_state(self, state: str) -> str:
    return f"{self.id}_{state}"   # game_object.lua getUniqueState


class UpdaterRegistry:
  """lua/modules/updater_registry.lua:60-303, restated.  Lua tables iterated
  with `pairs()` become insertion-ordered dicts (A11)."""

  def __init__(self):
    self._update_table: Dict[int, List[Dict[str, Any]]] = {}
    self._group_prefix = "__"       # updater_registry.lua __init__: default prefix
    self._updater_count = 0
    self._by_state_and_updater_name: Optional[Dict[str, Dict[str, bool]]] = None

  def register_updater(self, update_fn: Optional[Callable] = None, *, priority: int = 100,
                       start_frame: int = 0, probability: float = 1.0,
                       group: Optional[str] = None, state: Optional[str] = None,
                       states: Optional[List[str]] = None, _updater_name: Optional[str] = None,
                       tag: Optional[str] = None):
    """registerUpdater (:114-163).  `tag` (not in the reference) names the
    registration for `level_update_order`."""
    add_group = False
    if group is None:   # :147-152
      group = f"UPDATER_GRP__{self._group_prefix}_Updater#{self._updater_count}"
      self._updater_count += 1
      add_group = True
    self._update_table.setdefault(priority, []).append({
        "update_fn": update_fn, "start_frame": start_frame, "probability": probability,
        "group": group, "state": state, "states": states, "_updater_name": _updater_name,
        "_add_group": add_group, "tag": tag})

  def set_group_prefix(self, prefix: str):   # :165-168
    self._group_prefix = prefix
    self._updater_count = 0

  def get_sorted_priorities(self) -> List[int]:   # :170-177: descending
    return sorted(self._update_table, reverse=True)

  def uniquify_states_and_add_groups(self, game_object: GameObjectStates):   # :186-217
    self._by_state_and_updater_name = {}
    for priority, specs in self._update_table.items():
      for spec in specs:
        if spec["states"] is None:
          spec["states"] = (game_object.get_all_states() if spec["state"] is None
                            else [spec["state"]])
          spec["state"] = None
        spec["_updater_name"] = f"_priority_{priority}_{spec['group']}"
        for i, state in enumerate(spec["states"]):
          if spec["_add_group"]:
            game_object.get_groups_for_state(state).append(spec["group"])
          spec["states"][i] = game_object.get_unique_state(state)
          self._by_state_and_updater_name.setdefault(state, {})[spec["_updater_name"]] = True

  def merge_with(self, other: "UpdaterRegistry"):   # :220-258
    if self._by_state_and_updater_name is None:
      self._by_state_and_updater_name = {}
    seen = self._by_state_and_updater_name
    for priority, specs in other._update_table.items():
      for spec in specs:
        all_states = all(seen.get(state, {}).get(spec["_updater_name"]) for state in spec["states"])
        if all_states and spec["states"]:
          continue
        self.register_updater(spec["update_fn"], priority=priority,
                              start_frame=spec["start_frame"], probability=spec["probability"],
                              group=spec["group"], states=list(spec["states"]),
                              _updater_name=spec["_updater_name"], tag=spec["tag"])
        for state in spec["states"]:
          seen.setdefault(state, {})[spec["_updater_name"]] = True

  def add_update_order(self, update_order: List[str]):   # :261-273
    for priority in self.get_sorted_priorities():
      names: Dict[str, bool] = {}
      for spec in self._update_table[priority]:
        names[spec["_updater_name"]] = True
      update_order.extend(names)

  def specs(self, priority: int) -> List[Dict[str, Any]]:
    return self._update_table.get(priority, [])


# --------------------------------------------------------------------------
# registerUpdaters of the components on the hot path: (tag, kwargs) in source
# order.  kwargs reference the component's own kwargs through callables.
Reg = Tuple[str, Dict[str, Any]]


def _avatar(kw) -> List[Reg]:
  # avatar_library.lua:155-203: `move` (or moveAbsolute), priority 150,
  # probability = speed
  return [("Avatar.move", dict(priority=150, probability=float(kw.get("speed", 1.0))))]


def _zapper(kw) -> List[Reg]:
  # avatar_library.lua:633-649
  return [("Zapper.zap", dict(priority=140)),
          ("Zapper.respawn", dict(priority=135, state="<waitState>",
                                  start_frame=int(kw["framesTillRespawn"])))]


def _cleaner(kw) -> List[Reg]:
  # clean_up/components.lua:221-232
  return [("Cleaner.clean", dict(priority=140)),
          ("Cleaner.resetCumulant", dict(priority=400))]


def _taste(kw) -> List[Reg]:
  # clean_up/components.lua:431-434 (the Taste components of the other levels
  # register nothing: territory/components.lua, commons_harvest/components.lua)
  return [("Taste.resetCumulant", dict(priority=400))]


def _global_data(kw) -> List[Reg]:
  return [("GlobalData.resetCumulants", dict(priority=2))]   # :488-491


def _all_nonself_cumulants(kw) -> List[Reg]:
  # clean_up/components.lua:542-556
  return [("AllNonselfCumulants.getCumulants", dict(priority=4)),
          ("AllNonselfCumulants.resetCumulants", dict(priority=400))]


def _animation(kw) -> List[Reg]:
  # component_library.lua:1070-1094: one updater per state of the cycle, all in
  # the component's group, startFrame = gameFramesPerAnimationFrame
  return [(f"Animation.{state}", dict(state=state, group=kw.get("group"),
                                      start_frame=int(kw["gameFramesPerAnimationFrame"])))
          for state in kw["states"]]


def _stochastic_interval_episode_ending(kw) -> List[Reg]:
  # component_library.lua:936-939
  return [("StochasticIntervalEpisodeEnding.maybeEndEpisode",
           dict(start_frame=int(kw["minimumFramesPerEpisode"])))]


def _density_regrow(kw) -> List[Reg]:
  # commons_harvest/components.lua:104-137: one sprout updater per neighbour
  # count 0 .. upperBoundPossibleNeighbors - 1 (= floor(pi r^2) + 1, :84-85);
  # counts beyond the declared probabilities use the last one
  import math
  probs = list(kw["regrowthProbabilities"])
  upper = int(math.floor(math.pi * float(kw["radius"]) ** 2)) + 1
  return [(f"DensityRegrow.sprout_{k}",
           dict(priority=10, group=f"waits_{k}", state=f"appleWait_{k}",
                probability=float(probs[min(k, len(probs) - 1)])))
          for k in range(upper)]


def _resource(kw) -> List[Reg]:
  # territory/components.lua:97-117
  return [("Resource.provideRewards", dict(group="claimedResources",
                                           probability=float(kw["rewardRate"]),
                                           start_frame=int(kw["rewardDelay"]))),
          ("Resource.releaseClaimOfDeadAgent", dict(group="claimedResources", priority=2,
                                                    start_frame=5))]


def _resource_claimer(kw) -> List[Reg]:
  return [("ResourceClaimer.claim", dict())]   # territory/components.lua:273-275


def _paintbrush(kw) -> List[Reg]:
  return [("Paintbrush.drawBrush", dict(priority=130))]   # territory/components.lua:408-411


def _graduated_sanctions_marking(kw) -> List[Reg]:
  return [("GraduatedSanctionsMarking.resetToInitialLevel", dict(priority=3))]   # avatar_library.lua:1022-1025


def _choice_coin_regrow(kw) -> List[Reg]:
  # coins/components.lua:193-199
  return [("ChoiceCoinRegrow.regrow", dict(state="<waitState>",
                                           probability=float(kw["regrowRate"])))]


def _fixed_rate_regrow(kw) -> List[Reg]:
  # coop_mining/components.lua:45-60: one updater per live state, on the pieces in waitState
  return [(f"FixedRateRegrow.regrow_{i}",
           dict(priority=200, state=kw["waitState"], probability=float(rate)))
          for i, rate in enumerate(kw["liveRates"])]


def _gift_beam(kw) -> List[Reg]:
  return [("GiftBeam.gift", dict(priority=140))]   # gift_refinements/components.lua:186-211


def _matrix_resource(kw) -> List[Reg]:
  # the_matrix/components.lua:84-101 (the draw is the function's own)
  return [("Resource.maybeRespawn", dict(priority=100, state=kw["waitState"],
                                         start_frame=int(kw["regenerationDelay"])))]


def _spawn_resources_when_all_players_zapped(kw) -> List[Reg]:
  return [("SpawnResourcesWhenAllPlayersZapped.step", dict(priority=7))]   # :303-321


def _game_interaction_zapper(kw) -> List[Reg]:
  # the_matrix/components.lua:396-472, in source order
  return [("GameInteractionZapper.zap", dict(priority=140)),
          ("GameInteractionZapper.respawn", dict(priority=135, state="<waitState>",
                                                 start_frame=int(kw["framesTillRespawn"]))),
          ("GameInteractionZapper.applyScheduledEffects", dict(priority=4, state="<aliveState>")),
          ("GameInteractionZapper.endEpisodeIfApplicable", dict(priority=900)),
          ("GameInteractionZapper.resetSimultaneousInteractionBlocker", dict(priority=890))]


def _ready_to_interact_marker(kw) -> List[Reg]:
  return [("ReadyToInteractMarker.displayReadiness", dict(priority=2))]   # :1081-1096


COMPONENT_UPDATERS: Dict[str, Callable[[Mapping[str, Any]], List[Reg]]] = {
    "the_matrix/Resource": _matrix_resource,
    "SpawnResourcesWhenAllPlayersZapped": _spawn_resources_when_all_players_zapped,
    "GameInteractionZapper": _game_interaction_zapper,
    "ReadyToInteractMarker": _ready_to_interact_marker,
    "Avatar": _avatar,
    "Zapper": _zapper,
    "Cleaner": _cleaner,
    "clean_up/Taste": _taste,
    "GlobalData": _global_data,
    "AllNonselfCumulants": _all_nonself_cumulants,
    "Animation": _animation,
    "StochasticIntervalEpisodeEnding": _stochastic_interval_episode_ending,
    "DensityRegrow": _density_regrow,
    "Resource": _resource,
    "ResourceClaimer": _resource_claimer,
    "Paintbrush": _paintbrush,
    "GraduatedSanctionsMarking": _graduated_sanctions_marking,
    "ChoiceCoinRegrow": _choice_coin_regrow,
    # (coop_mining's Ore, MineBeam and MiningTracker register none: the beam leaves from
    # MineBeam:update, components.lua:228-244)
    "FixedRateRegrow": _fixed_rate_regrow,
    # gift_refinements' FixedRateRegrow is a component update() (components.lua:45-55): no updater
    "gift_refinements/FixedRateRegrow": lambda kw: [],
    "GiftBeam": _gift_beam,
    # collaborative_cooking/components.lua:79-100, 165-181, 452-470, 495-512
    "InteractBeam": lambda kw: [("InteractBeam.interact", dict(priority=140))],
    "Container": lambda kw: [("Container.tick", dict(priority=140))],
    "CookingPot": lambda kw: [("CookingPot.tickPotFn", dict(priority=140))],
    "LoadingBarVisualiser": lambda kw: [("LoadingBarVisualiser.tickLoadingBarFn", dict(priority=140))],
    # externality_mushrooms/components.lua:164-182, 321-335 (one updater per live state, in the
    # table's order: pairs()), 361-370; MushroomEating, MushroomRegrowth and Destroyable register none
    "MushroomGrowable": lambda kw: [("MushroomGrowable.registration", dict(priority=500))],
    "Perishable": lambda kw: [(f"Perishable.perish_{i}", dict(priority=3, state=state, start_frame=int(delay)))
                              for i, (state, delay) in enumerate(kw["delayPerState"].items())],
    "Cumulants": lambda kw: [("Cumulants.resetCumulants", dict(priority=900))],
}


def level_update_order(objects: Sequence[Mapping[str, Any]],
                       level: str = "") -> List[Tuple[int, str]]:
  """The frame's schedule of a level: [(priority, tag)], priority descending,
  registrations of one priority in first-registration order (A11), one entry
  per distinct (priority, tag).  `objects`: the level's game-object configs in
  creation order (scene, avatars, then the map's objects row-major,
  base_simulation.lua:103-131); `level`: the Lua level (settings["levelName"]),
  for components that exist per level under one name."""
  merged: Dict[int, List[str]] = {}
  for obj in objects:
    for comp in obj["components"]:
      fn = (COMPONENT_UPDATERS.get(f"{level}/{comp['component']}") or
            COMPONENT_UPDATERS.get(comp["component"]))
      if fn is None:
        continue
      for tag, kw in fn(comp.get("kwargs", {}) or {}):
        bucket = merged.setdefault(int(kw.get("priority", 100)), [])
        if tag not in bucket:
          bucket.append(tag)
  return [(p, tag) for p in sorted(merged, reverse=True) for tag in merged[p]]
