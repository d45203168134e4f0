"""Host-side mirror of the reference's public substrate API for the hot path.

Same names, argument meaning and error behaviour as
  meltingpot/substrate.py:41-113           SUBSTRATES, get_config, build
  meltingpot/utils/substrates/substrate.py:50-104   Substrate.reset/step/
      observation_spec/action_spec/reward_spec/discount_spec/close
so that a training loop written against `meltingpot.substrate.build(name,
roles=...)` can switch to `meltingpot_amd.substrate.build(name, roles=...,
num_worlds=N)`.  Everything that computes lives in libmp_engine.so; this module
only validates arguments, owns the observation tensors and shapes the returned
timesteps.  (`dm_env` is not a dependency: `TimeStep` / `StepType` / the spec
classes below duck-type it — same field names, `.first()/.mid()/.last()`,
`spec.validate()`, `spec.minimum/.maximum/.num_values`.)

Two result shapes:
  * num_worlds == 1 (default), batched=False: exactly the reference's —
    `TimeStep(step_type, reward=[P x float64], discount=float,
    observation=[P x {"RGB", "READY_TO_SHOOT",
    "NUM_OTHERS_WHO_CLEANED_THIS_STEP", "COLLECTIVE_REWARD", "WORLD.RGB"}])` with
    numpy leaves (clean_up.py:813-832, collective_reward_wrapper.py:25);
  * batched: every leaf gains a leading [N] axis and is a torch tensor that
    lives on the GPU the engine runs on (no host copy, no sync).
"""

from __future__ import annotations

import dataclasses
import enum
import os
from typing import Any, Dict, List, Mapping, NamedTuple, Optional, Sequence, Tuple

import numpy as np

from meltingpot_amd import engine as engine_lib

# --------------------------------------------------------------------------
# dm_env duck types


class StepType(enum.IntEnum):
  FIRST = 0
  MID = 1
  LAST = 2

  def first(self) -> bool:
    return self is StepType.FIRST

  def mid(self) -> bool:
    return self is StepType.MID

  def last(self) -> bool:
    return self is StepType.LAST


class TimeStep(NamedTuple):
  step_type: Any
  reward: Any
  discount: Any
  observation: Any

  def first(self) -> bool:
    return self.step_type == StepType.FIRST

  def mid(self) -> bool:
    return self.step_type == StepType.MID

  def last(self) -> bool:
    return self.step_type == StepType.LAST


class RolloutTimeStep(TimeStep):
  """A TimeStep of a substrate built with `rollout_length=T`: the same four fields
  (it IS a TimeStep: unpacking, `_replace`, `.first()` all work) plus `.slot`, the
  index of this step along the leading axis of `Substrate.rollout`'s [T, N, ...]
  tensors.  Its leaves are views of that slot: they stay valid until the substrate has
  been stepped T more times."""
  slot: int = -1

  def _replace(self, **kwargs):   # (a NamedTuple's _replace builds a fresh tuple: carry the slot)
    out = super()._replace(**kwargs)
    out.slot = self.slot
    return out


class Array:
  """dm_env.specs.Array look-alike."""

  def __init__(self, shape, dtype, name=None):
    self.shape = tuple(shape)
    self.dtype = np.dtype(dtype)
    self.name = name

  def validate(self, value):
    value = np.asarray(value)
    if value.shape != self.shape:
      raise ValueError(f"{self.name}: shape {value.shape} != {self.shape}")
    if value.dtype != self.dtype:
      raise ValueError(f"{self.name}: dtype {value.dtype} != {self.dtype}")
    return value

  def replace(self, **kw):
    out = self.__class__.__new__(self.__class__)
    out.__dict__.update(self.__dict__)
    out.__dict__.update(kw)
    return out

  def __repr__(self):
    return f"{type(self).__name__}(shape={self.shape}, dtype={self.dtype}, name={self.name!r})"

  # dm_env specs compare by value — shape and dtype (and bounds), NOT the name
  # (dm_env/specs.py Array.__eq__, BoundedArray.__eq__): the reference's
  # discrete_action_wrapper.py:91 compares the players' action specs with `!=`, its
  # substrate_test.py:36-47 an env's specs against the factory's differently named ones
  def __eq__(self, other):
    return isinstance(other, Array) and self.shape == other.shape and self.dtype == other.dtype

  def __ne__(self, other):
    return not self == other

  def __hash__(self):
    return hash((self.shape, self.dtype))


class BoundedArray(Array):

  def __init__(self, shape, dtype, minimum, maximum, name=None):
    super().__init__(shape, dtype, name)
    self.minimum = np.asarray(minimum, self.dtype)
    self.maximum = np.asarray(maximum, self.dtype)

  def validate(self, value):
    value = super().validate(value)
    if (value < self.minimum).any() or (value > self.maximum).any():
      raise ValueError(f"{self.name}: value out of bounds")
    return value

  def __eq__(self, other):
    return (isinstance(other, BoundedArray) and Array.__eq__(self, other) and
            bool((self.minimum == other.minimum).all()) and bool((self.maximum == other.maximum).all()))

  __hash__ = Array.__hash__


class DiscreteArray(BoundedArray):
  """specs.action(n) (reference utils/substrates/specs.py:44-45): int64 scalar."""

  def __init__(self, num_values, dtype=np.int64, name="action"):
    super().__init__((), dtype, 0, num_values - 1, name)
    self.num_values = num_values


# --------------------------------------------------------------------------
# per-substrate configuration (the fields of the reference ConfigDict that the
# hot path needs; reference: configs/substrates/clean_up.py:806-838)


class SubstrateConfig:

  def __init__(self, name, action_set, individual_observation_names,
               global_observation_names, timestep_spec, valid_roles,
               default_player_roles, aux0_name, per_role_constants=False):
    self.name = name
    self.action_set = action_set
    self.individual_observation_names = list(individual_observation_names)
    self.global_observation_names = list(global_observation_names)
    self.action_spec = DiscreteArray(len(action_set))
    self.timestep_spec = dict(timestep_spec)
    self.valid_roles = frozenset(valid_roles)
    self.default_player_roles = tuple(default_player_roles)
    self.aux0_name = aux0_name
    self.per_role_constants = per_role_constants

  # the reference hands out a locked ml_collections.ConfigDict (substrate.py:41-55);
  # callers written against it say `with config.unlocked(): config.x = ...`
  def lock(self):
    return self

  def unlock(self):
    return self

  def unlocked(self):
    import contextlib
    return contextlib.nullcontext(self)


_NOOP = {"move": 0, "turn": 0, "fireZap": 0, "fireClean": 0}


def _clean_up_config() -> SubstrateConfig:
  # clean_up.py:461-483 (ACTION_SET order is what the discrete ids index)
  def a(**kw):
    d = dict(_NOOP)
    d.update(kw)
    return d
  action_set = (a(), a(move=1), a(move=3), a(move=4), a(move=2), a(turn=-1),
                a(turn=1), a(fireZap=1), a(fireClean=1))
  return SubstrateConfig(
      name="clean_up",
      action_set=action_set,
      individual_observation_names=("RGB", "READY_TO_SHOOT",
                                    "NUM_OTHERS_WHO_CLEANED_THIS_STEP"),
      global_observation_names=("WORLD.RGB",),
      timestep_spec={
          "RGB": Array((88, 88, 3), np.uint8, "RGB"),
          "READY_TO_SHOOT": Array((), np.float64, "READY_TO_SHOOT"),
          "NUM_OTHERS_WHO_CLEANED_THIS_STEP": Array(
              (), np.float64, "NUM_OTHERS_WHO_CLEANED_THIS_STEP"),
          "WORLD.RGB": Array((168, 240, 3), np.uint8, "WORLD.RGB"),
      },
      valid_roles={"default"},
      default_player_roles=("default",) * 7,
      aux0_name="NUM_OTHERS_WHO_CLEANED_THIS_STEP")


def _commons_harvest_config(name: str, players: int) -> SubstrateConfig:
  # commons_harvest__open.py:252-273 (ACTION_SET), :531-558 (get_config); the
  # __closed variant shares the Lua level, the action set and the specs.  The
  # __open pack is lowered for 16 players (BASELINE.json configs[2]; the
  # reference's default is 7), the __closed pack for the default 7.
  def a(**kw):
    d = {"move": 0, "turn": 0, "fireZap": 0}
    d.update(kw)
    return d
  action_set = (a(), a(move=1), a(move=3), a(move=4), a(move=2), a(turn=-1),
                a(turn=1), a(fireZap=1))
  return SubstrateConfig(
      name=name,
      action_set=action_set,
      individual_observation_names=("RGB", "READY_TO_SHOOT"),
      global_observation_names=("WORLD.RGB",),
      timestep_spec={
          "RGB": Array((88, 88, 3), np.uint8, "RGB"),
          "READY_TO_SHOOT": Array((), np.float64, "READY_TO_SHOOT"),
          "WORLD.RGB": Array((144, 192, 3), np.uint8, "WORLD.RGB"),
      },
      valid_roles={"default"},
      default_player_roles=("default",) * players,
      aux0_name=None)


def _territory_config(name: str, world_hw, players: int = 9) -> SubstrateConfig:
  # territory.py:578-602 (ACTION_SET), territory__rooms.py:84-104 /
  # territory__open.py:111-131 (get_config)
  def a(**kw):
    d = {"move": 0, "turn": 0, "fireZap": 0, "fireClaim": 0}
    d.update(kw)
    return d
  action_set = (a(), a(move=1), a(move=3), a(move=4), a(move=2), a(turn=-1),
                a(turn=1), a(fireZap=1), a(fireClaim=1))
  return SubstrateConfig(
      name=name,
      action_set=action_set,
      individual_observation_names=("RGB", "READY_TO_SHOOT"),
      global_observation_names=("WORLD.RGB",),
      timestep_spec={
          "RGB": Array((88, 88, 3), np.uint8, "RGB"),
          "READY_TO_SHOOT": Array((), np.float64, "READY_TO_SHOOT"),
          "WORLD.RGB": Array(tuple(world_hw) + (3,), np.uint8, "WORLD.RGB"),
      },
      valid_roles={"default"},
      default_player_roles=("default",) * players,
      aux0_name=None)


def _coins_config() -> SubstrateConfig:
  # coins.py:431-491 (ACTION_SET: move / turn only, get_config)
  def a(move=0, turn=0):
    return {"move": move, "turn": turn}
  action_set = (a(), a(move=1), a(move=3), a(move=4), a(move=2), a(turn=-1), a(turn=1))
  return SubstrateConfig(
      name="coins",
      action_set=action_set,
      individual_observation_names=("RGB", "MISMATCHED_COIN_COLLECTED_BY_PARTNER"),
      global_observation_names=("WORLD.RGB",),
      timestep_spec={
          "RGB": Array((88, 88, 3), np.uint8, "RGB"),
          "MISMATCHED_COIN_COLLECTED_BY_PARTNER": Array(
              (), np.float64, "MISMATCHED_COIN_COLLECTED_BY_PARTNER"),
          "WORLD.RGB": Array((136, 136, 3), np.uint8, "WORLD.RGB"),
      },
      valid_roles={"default"},
      default_player_roles=("default",) * 2,
      aux0_name="MISMATCHED_COIN_COLLECTED_BY_PARTNER")


def _coop_mining_config() -> SubstrateConfig:
  # coop_mining.py:423-473 (ACTION_SET: move / turn / mine; get_config)
  def a(**kw):
    d = {"move": 0, "turn": 0, "mine": 0}
    d.update(kw)
    return d
  action_set = (a(), a(move=1), a(move=3), a(move=4), a(move=2), a(turn=-1), a(turn=1),
                a(mine=1))
  return SubstrateConfig(
      name="coop_mining",
      action_set=action_set,
      individual_observation_names=("RGB", "READY_TO_SHOOT"),
      global_observation_names=("WORLD.RGB",),
      timestep_spec={
          "RGB": Array((88, 88, 3), np.uint8, "RGB"),
          "READY_TO_SHOOT": Array((), np.float64, "READY_TO_SHOOT"),
          "WORLD.RGB": Array((216, 216, 3), np.uint8, "WORLD.RGB"),
      },
      # (both roles build the same avatar: MineBeam.agentRole is "none" for all)
      valid_roles={"default", "target"},
      default_player_roles=("default",) * 6,
      aux0_name=None)


def _gift_refinements_config() -> SubstrateConfig:
  # gift_refinements.py:410-477 (ACTION_SET: move / turn / refineAndGift / consumeTokens; get_config)
  def a(**kw):
    d = {"move": 0, "turn": 0, "refineAndGift": 0, "consumeTokens": 0}
    d.update(kw)
    return d
  action_set = (a(), a(move=1), a(move=3), a(move=4), a(move=2), a(turn=-1), a(turn=1),
                a(refineAndGift=1), a(consumeTokens=1))
  return SubstrateConfig(
      name="gift_refinements",
      action_set=action_set,
      individual_observation_names=("RGB", "READY_TO_SHOOT", "INVENTORY"),
      global_observation_names=("WORLD.RGB",),
      timestep_spec={
          "RGB": Array((88, 88, 3), np.uint8, "RGB"),
          "READY_TO_SHOOT": Array((), np.float64, "READY_TO_SHOOT"),
          "INVENTORY": Array((3,), np.float64, "INVENTORY"),
          "WORLD.RGB": Array((216, 216, 3), np.uint8, "WORLD.RGB"),
      },
      # (both roles build the same avatar: GiftBeam.agentRole is "none" for all)
      valid_roles={"default", "target"},
      default_player_roles=("default",) * 6,
      aux0_name=None)


def _externality_mushrooms_config(name: str) -> SubstrateConfig:
  # externality_mushrooms.py:633-658 (ACTION_SET), :1027-1048 (get_config);
  # externality_mushrooms__dense.py:69-86 (the map, the specs, five players)
  def a(**kw):
    d = {"move": 0, "turn": 0, "fireZap": 0}
    d.update(kw)
    return d
  action_set = (a(), a(move=1), a(move=3), a(move=4), a(move=2), a(turn=-1), a(turn=1),
                a(fireZap=1))
  return SubstrateConfig(
      name=name,
      action_set=action_set,
      individual_observation_names=("RGB", "READY_TO_SHOOT"),
      global_observation_names=("WORLD.RGB",),
      timestep_spec={
          "RGB": Array((88, 88, 3), np.uint8, "RGB"),
          "READY_TO_SHOOT": Array((), np.float64, "READY_TO_SHOOT"),
          "WORLD.RGB": Array((112, 184, 3), np.uint8, "WORLD.RGB"),
      },
      valid_roles={"default"},
      default_player_roles=("default",) * 5,
      aux0_name=None)


# layout -> (WORLD.RGB height, width, default players): collaborative_cooking__<layout>.py
_COOKING_LAYOUTS = {"asymmetric": (40, 72, 2), "circuit": (40, 72, 2), "cramped": (40, 72, 2),
                    "crowded": (72, 104, 9), "figure_eight": (72, 128, 6), "forced": (40, 72, 2),
                    "ring": (40, 72, 2)}


def _cooking_config(layout: str) -> SubstrateConfig:
  # collaborative_cooking.py:696-724 (ACTION_SET), :899-921 (get_config) + the layout module
  def a(**kw):
    d = {"move": 0, "turn": 0, "interact": 0}
    d.update(kw)
    return d
  action_set = (a(), a(move=1), a(move=3), a(move=4), a(move=2), a(turn=-1), a(turn=1),
                a(interact=1))
  h, w, players = _COOKING_LAYOUTS[layout]
  return SubstrateConfig(
      name=f"collaborative_cooking__{layout}",
      action_set=action_set,
      individual_observation_names=("RGB",),
      global_observation_names=("WORLD.RGB",),
      timestep_spec={
          "RGB": Array((40, 40, 3), np.uint8, "RGB"),
          "WORLD.RGB": Array((h, w, 3), np.uint8, "WORLD.RGB"),
      },
      valid_roles={"default"},
      default_player_roles=("default",) * players,
      aux0_name=None)


def _matrix_config(name: str, resources: int, arena: bool, roles, valid_roles) -> SubstrateConfig:
  # prisoners_dilemma_in_the_matrix__repeated.py:153-173 (ACTION_SET, shared by all
  # fifteen), :518-552 (get_config); arenas: 8 players, 11 x 11 window, 24 x 25 map
  # (prisoners_dilemma_in_the_matrix__arena.py:473-512); repeated / one_shot: 2
  # players, 5 x 5 window, 15 x 23 map
  def a(**kw):
    d = {"move": 0, "turn": 0, "interact": 0}
    d.update(kw)
    return d
  action_set = (a(), a(move=1), a(move=3), a(move=4), a(move=2), a(turn=-1),
                a(turn=1), a(interact=1))
  rgb = (88, 88, 3) if arena else (40, 40, 3)
  world = (192, 200, 3) if arena else (120, 184, 3)
  return SubstrateConfig(
      name=name,
      action_set=action_set,
      individual_observation_names=("RGB", "INVENTORY", "READY_TO_SHOOT",
                                    "INTERACTION_INVENTORIES"),
      global_observation_names=("WORLD.RGB",),
      timestep_spec={
          "RGB": Array(rgb, np.uint8, "RGB"),
          "INVENTORY": Array((resources,), np.float64, "INVENTORY"),
          "READY_TO_SHOOT": Array((), np.float64, "READY_TO_SHOOT"),
          "INTERACTION_INVENTORIES": Array((2, resources), np.float64,
                                           "INTERACTION_INVENTORIES"),
          "WORLD.RGB": Array(world, np.uint8, "WORLD.RGB"),
      },
      valid_roles=set(valid_roles),
      default_player_roles=tuple(roles),
      aux0_name=None,
      # the per-player constants of the roles (DyadicRole, avatar colour) are in
      # the pack per (role, player): engine.pack_role_names
      per_role_constants=len(valid_roles) > 1)


def _matrix_configs():
  out = {}
  three = {"pure_coordination", "rationalizable_coordination", "running_with_scissors"}
  for game in ("prisoners_dilemma", "chicken", "stag_hunt", "pure_coordination",
               "rationalizable_coordination", "bach_or_stravinsky", "running_with_scissors"):
    variants = ("repeated", "arena") + (("one_shot",) if game == "running_with_scissors" else ())
    for variant in variants:
      name = f"{game}_in_the_matrix__{variant}"
      arena = variant == "arena"
      if game == "bach_or_stravinsky":
        # bach_or_stravinsky_in_the_matrix__repeated.py:535-536, __arena.py:537-538
        roles = (("bach_fan",) * 4 + ("stravinsky_fan",) * 4) if arena else (
            "bach_fan", "stravinsky_fan")
        valid = {"default", "bach_fan", "stravinsky_fan"}
      else:
        roles = ("default",) * (8 if arena else 2)
        valid = {"default"}
      out[name] = (lambda n=name, g=game, ar=arena, r=roles, v=valid:
                   _matrix_config(n, 3 if g in three else 2, ar, r, v))
  return out


_CONFIGS = {
    **_matrix_configs(),
    "coins": _coins_config,
    "coop_mining": _coop_mining_config,
    "gift_refinements": _gift_refinements_config,
    **{f"collaborative_cooking__{_l}": (lambda _l=_l: _cooking_config(_l)) for _l in _COOKING_LAYOUTS},
    "externality_mushrooms__dense": lambda: _externality_mushrooms_config("externality_mushrooms__dense"),
    "territory__rooms": lambda: _territory_config("territory__rooms", (168, 168)),
    "territory__open": lambda: _territory_config("territory__open", (184, 312)),
    "territory__inside_out": lambda: _territory_config("territory__inside_out", (184, 184), 5),
    "clean_up": _clean_up_config,
    "commons_harvest__open": lambda: _commons_harvest_config("commons_harvest__open", 7),
    "commons_harvest__closed": lambda: _commons_harvest_config("commons_harvest__closed", 7),
    "commons_harvest__partnership": lambda: _commons_harvest_config(
        "commons_harvest__partnership", 7),
}
SUBSTRATES = frozenset(_CONFIGS)


def get_config(name: str) -> SubstrateConfig:
  """reference: meltingpot/substrate.py:41-55."""
  if name not in SUBSTRATES:
    raise ValueError(f"{name} not in {sorted(SUBSTRATES)} (substrates with a "
                     "HIP engine in this build).")
  return _CONFIGS[name]()


# --------------------------------------------------------------------------


class Subject:
  """The slice of `reactivex.subject.Subject` the reference's Substrate uses
  (utils/substrates/substrate.py:56-104): subscribe / on_next / on_completed.
  reactivex is not a dependency of this package."""

  def __init__(self):
    self._observers = []
    self._completed = False

  def subscribe(self, on_next=None, on_error=None, on_completed=None):
    if on_next is not None and not callable(on_next) and hasattr(on_next, "on_next"):
      # an observer object (reactivex accepts one in place of the three callbacks:
      # utils/evaluation/evaluation.py:88-99 subscribes a ReturnSubject this way)
      observer = on_next
      on_next = observer.on_next
      on_error = getattr(observer, "on_error", None)
      on_completed = getattr(observer, "on_completed", None)
    obs = (on_next, on_error, on_completed)
    if self._completed:
      if on_completed:
        on_completed()
    else:
      self._observers.append(obs)
    subject = self

    class _Disposable:
      def dispose(self):
        if obs in subject._observers:
          subject._observers.remove(obs)
    return _Disposable()

  def on_next(self, value):
    for on_next, _, _ in list(self._observers):
      if on_next:
        on_next(value)

  def on_error(self, error):
    for _, on_error, _ in list(self._observers):
      if on_error:
        on_error(error)

  def pipe(self, *operators):
    """reactivex's `Observable.pipe`: each operator maps an observable to an observable."""
    out = self
    for op in operators:
      out = op(out)
    return out

  def on_completed(self):
    self._completed = True
    observers, self._observers = self._observers, []
    for _, _, on_completed in observers:
      if on_completed:
        on_completed()


@dataclasses.dataclass(frozen=True)
class SubstrateObservables:
  """substrate.py:30-45: `action`, `timestep` and `events` streams (the `dmlab2d`
  member has no counterpart: there is no dmlab2d underneath)."""
  action: Subject
  timestep: Subject
  events: Subject
  # batched substrates only: (world, (name, payload)) for every world of the batch
  # (`events` stays reference-shaped: the events of world 0)
  events_batched: Optional[Subject] = None


def resolve_env_seed(env_seed: Optional[int]) -> int:
  """builder.py:174-176: `if env_seed is None: env_seed = <random seed>`.  World w
  of a batch is seeded env_seed + w (one env_seed per world, as N reference
  environments built with consecutive seeds).  Any int is a seed — 0 and negative
  ones included — taken modulo 2**64 (the engine's seeds are u64)."""
  if env_seed is None:
    env_seed = int.from_bytes(os.urandom(8), "little") >> 1
  return int(env_seed) % (1 << 64)


def action_fields(eng) -> Tuple[Tuple[str, ...], Tuple[Tuple[int, int, int], ...]]:
  """(names, (min, max, default) per name) of the avatars' raw action fields in
  actionOrder — the pack's "action_names" / "action_spec" tables
  (avatar_library.lua:205-223 Avatar:discreteActionSpec)."""
  from meltingpot_amd import pack as pack_lib
  t = eng.pack_tables() if hasattr(eng, "pack_tables") else pack_lib.loads(eng.pack_bytes)
  names = tuple(n.decode() for n in bytes(t["action_names"]).split(b"\0")[:-1])
  spec = tuple(tuple(int(v) for v in row) for row in t["action_spec"].reshape(-1, 3))
  assert len(names) == len(spec)
  return names, spec


def validate_action_table(action_table, names, ranges) -> np.ndarray:
  """discrete_action_wrapper.py:28-49: every row names exactly the action spec's
  fields with values inside their ranges.  Returns the table as int32 [K, A]."""
  if not action_table:
    raise ValueError("action_table must not be empty")
  rows = np.zeros((len(action_table), len(names)), np.int32)
  for i, action in enumerate(action_table):
    ok = set(action) == set(names)
    if ok:
      for a, (n, (lo, hi, _)) in enumerate(zip(names, ranges)):
        v = int(action[n])
        ok = ok and lo <= v <= hi
        rows[i, a] = v
    if not ok:
      raise ValueError(f"Action {i} ({dict(action)}) does not match action_spec "
                       f"({dict(zip(names, ranges))}).")
  return rows


class Substrate:
  """N worlds of one substrate behind the reference's `Substrate` interface.

  In batched mode the leaves of every TimeStep are the SAME device tensors,
  refreshed in place by the next reset() / step(): clone what you keep — or build
  with `rollout_length=T` and keep them for free: every leaf is then a slot of a
  [T, N, ...] ring the engine writes in turn (mp_bind_output_ring), a TimeStep's
  leaves stay untouched for T steps, and `Substrate.rollout` is the whole ring."""

  def __init__(self, config: SubstrateConfig, roles: Sequence[str],
               pack_bytes: bytes, *, num_worlds: int = 1, batched: Optional[bool] = None,
               device: int = 0, env_seed: Optional[int] = None,
               auto_reset: bool = True, world_offset: int = 0,
               debug_observations: bool = False,
               action_table: Optional[Sequence[Mapping[str, int]]] = None,
               rollout_length: int = 0, check_device_actions: bool = False):
    """`action_table`: the discrete actions, as in the reference's
    `build_substrate(..., action_table)` (utils/substrates/substrate.py:107-139,
    discrete_action_wrapper.py:77-109): row i is what discrete action i does,
    any combination of the avatar's raw fields.  Default: the config's
    ACTION_SET (looked up on the device by mp_step).

    `check_device_actions`: a debugging switch — device action tensors are range-checked
    like host arrays are (ValueError; costs a host synchronisation per step).  Off, ids
    outside the table do NOOP and are counted (mp_counters: bad_actions).

    `rollout_length` = T > 0 (batched substrates): the observations a learner keeps
    without a copy.  The reference hands back fresh arrays every step
    (wrappers/multiplayer_wrapper.py:108-118, substrate.py:74-81) and a rollout just
    stores them; here submission t (every reset() and step()) writes slot t % T of
    [T, N, ...] tensors, the returned `RolloutTimeStep` holds views of its slot
    (`.slot`), and nothing is cloned, synchronised or re-tuned between steps."""
    invalid = set(roles) - config.valid_roles  # configs/substrates/__init__.py:42-45
    if invalid:
      raise ValueError(f"Invalid roles: {invalid!r}. Must be one of "
                       f"{config.valid_roles!r}")
    if num_worlds < 1:
      raise ValueError("num_worlds must be positive")
    self._config = config
    self._roles = tuple(roles)
    self._batched = (num_worlds > 1) if batched is None else bool(batched)
    if not self._batched and num_worlds != 1:
      raise ValueError("batched=False needs num_worlds == 1")
    self._T = int(rollout_length or 0)
    if self._T < 0:
      raise ValueError("rollout_length must not be negative")
    if self._T and not self._batched:
      raise ValueError("rollout_length needs a batched substrate (device tensors); the "
                       "one-world form already returns fresh numpy arrays every step")
    self._submissions = 0
    self._check_device_actions = bool(check_device_actions)
    if not self._roles:
      raise ValueError("roles must not be empty")
    # a config with several valid roles builds per-player constants from them
    # (bach_or_stravinsky_in_the_matrix__repeated.py:473-497): the pack carries
    # them per (role, player) and the engine is created for this assignment
    role_names = engine_lib.pack_role_names(pack_bytes) if config.per_role_constants else None
    role_ids = [role_names.index(r) for r in self._roles] if role_names else None
    env_seed = resolve_env_seed(env_seed)
    # num_players = len(roles) (configs/substrates/clean_up.py:847): the first
    # len(roles) avatars of the committed pack play
    self._eng = engine_lib.Engine(
        pack_bytes, num_worlds, device=device, auto_reset=auto_reset,
        world_offset=world_offset, base_seed=env_seed, literal_seed=True,
        num_players=len(self._roles),
        debug_observations=debug_observations, roles=role_ids)
    self._env_seed = env_seed
    self._action_rows = self._action_rows_dev = None
    if action_table is not None:
      names, ranges = action_fields(self._eng)
      self._action_rows = validate_action_table(action_table, names, ranges)
    E = engine_lib
    self._kinds = {"RGB": E.OBS_RGB, "WORLD.RGB": E.OBS_WORLD_RGB,
                   "READY_TO_SHOOT": E.OBS_READY_TO_SHOOT,
                   "COLLECTIVE_REWARD": E.OBS_COLLECTIVE_REWARD,
                   "INVENTORY": E.OBS_INVENTORY,
                   "INTERACTION_INVENTORIES": E.OBS_INTERACTION_INVENTORIES,
                   "POSITION": E.OBS_POSITION, "ORIENTATION": E.OBS_ORIENTATION}
    if config.aux0_name:
      self._kinds[config.aux0_name] = E.OBS_AUX0
    if config.name.split("__")[0] == "clean_up":
      # the debug observations a config built with _ENABLE_DEBUG_OBSERVATIONS reports
      # (clean_up.py:751-784): produced while bound, i.e. when a caller asks for them
      self._kinds.update({"PLAYER_CLEANED": E.OBS_AUX1, "PLAYER_ATE_APPLE": E.OBS_AUX2,
                          "NUM_OTHERS_PLAYER_ZAPPED_THIS_STEP": E.OBS_AUX3,
                          "NUM_OTHERS_WHO_ATE_THIS_STEP": E.OBS_AUX4})
    names = (list(config.individual_observation_names) +
             list(config.global_observation_names) + ["COLLECTIVE_REWARD"])
    unknown = [n for n in names if n not in self._kinds]
    if unknown:
      raise ValueError(f"observations {unknown} are not produced by the engine for "
                       f"{config.name!r} (it offers {sorted(self._kinds)})")
    kinds = {n: self._kinds[n] for n in names}
    kinds.update({"#reward": E.OBS_REWARD, "#discount": E.OBS_DISCOUNT,
                  "#step_type": E.OBS_STEP_TYPE})
    if self._batched and self._T:
      # the rollout ring: one [T, N, ...] tensor per leaf, written slot by slot
      bound = {n: self._eng.bind_ring(k, slots=self._T) for n, k in kinds.items()}
      self._host = None
    elif self._batched:
      bound = {n: self._eng.bind(k) for n, k in kinds.items()}
      self._host = None
    else:
      # one world, numpy leaves: every output lives in one device buffer that is
      # mirrored to pinned host memory with a single copy per step
      t = self._eng._torch
      layout, total = {}, 0
      for n, k in kinds.items():
        shape, dtype = self._eng.shapes[k]
        nbytes = int(np.prod(shape)) * t.empty((), dtype=dtype).element_size()
        layout[n] = (total, nbytes, shape, dtype)
        total += (nbytes + 255) & ~255
      self._blob = t.empty(total, dtype=t.uint8, device=self._eng.device)
      self._host = t.empty(total, dtype=t.uint8, pin_memory=self._blob.is_cuda)
      view = lambda buf, n: buf[layout[n][0]:layout[n][0] + layout[n][1]].view(
          layout[n][3]).view(layout[n][2])
      bound = {n: self._eng.bind(kinds[n], view(self._blob, n)) for n in kinds}
      self._host_views = {n: view(self._host, n).numpy() for n in kinds}
    self._obs = {n: bound[n] for n in names}
    self._reward = bound["#reward"]
    self._discount = bound["#discount"]
    self._step_type = bound["#step_type"]
    self._closed = False
    self._observables = SubstrateObservables(Subject(), Subject(), Subject(),
                                             Subject() if self._batched else None)

  # -- reference surface ---------------------------------------------------
  @property
  def num_worlds(self) -> int:
    return self._eng.N

  @property
  def num_players(self) -> int:
    return self._eng.P

  @property
  def engine(self) -> engine_lib.Engine:
    return self._eng

  @property
  def rollout(self) -> Optional[Dict[str, Any]]:
    """rollout_length=T: the ring itself — {"step_type": [T, N], "reward": [T, N, P],
    "discount": [T, N], "observation": {name: [T, N, ...]}} device tensors; slot s of
    every tensor is the same step (`RolloutTimeStep.slot`).  None otherwise."""
    if not self._T:
      return None
    return {"step_type": self._step_type, "reward": self._reward,
            "discount": self._discount, "observation": dict(self._obs)}

  @property
  def slot(self) -> int:
    """rollout_length=T: the slot the last reset() / step() wrote (-1 before the first) — the
    ENGINE's ring position (MpInfo.ring_next), so that submissions made through `.engine`
    (a masked reset, direct steps) are counted like this object's own."""
    if not self._T:
      return -1
    ring = self._eng.ring
    if not self._submissions and ring["next"] == 0:
      return -1
    return ring["last"]

  def reset(self) -> TimeStep:
    """Substrate.reset (substrate.py:66-72): FIRST, zero rewards, discount 0."""
    self._eng.use_current_stream()   # follow the caller's torch stream (ordered after the old one)
    self._eng.reset()
    self._submissions += 1
    return self._emit(self._timestep())

  def observation(self):
    """wrappers/base.py:60-62 `observation()`: the observation of the last reset() /
    step() again (the leaves of the last TimeStep)."""
    return self._timestep().observation

  # dmlab2d properties (wrappers/base.py:64-84): Melting Pot's levels register none —
  # the calls exist and answer like dmlab2d does for an unknown key
  def list_property(self, key: str = ""):
    if key:
      raise KeyError(key)
    return []

  def read_property(self, key: str):
    raise KeyError(key)

  def write_property(self, key: str, value):
    raise KeyError(key)

  def step(self, action) -> TimeStep:
    """Substrate.step (substrate.py:74-81).  `action`: P ints (unbatched), or
    an int tensor / array [N, P] (batched)."""
    t = self._eng._torch
    if self._batched:
      if isinstance(action, t.Tensor) and action.is_cuda:
        a = action.to(t.int32).contiguous()
        if self._check_device_actions:
          K = (len(self._action_rows) if self._action_rows is not None
               else self._eng.num_actions)
          if bool(((a < 0) | (a >= K)).any()):
            raise ValueError(f"actions must be in [0, {K})")
      else:
        a = np.asarray(action)
    else:
      a = np.asarray(action)
      if a.shape != (self._eng.P,):
        raise ValueError(f"Expected {self._eng.P} actions, got shape {a.shape}")
      a = a.reshape(1, self._eng.P)
    self._observables.action.on_next(action)
    self._eng.use_current_stream()
    if self._action_rows is None:
      self._eng.step(a)
    else:
      # a custom table: its rows go to the engine as raw fields (mp_step_fields)
      K = len(self._action_rows)
      if isinstance(a, t.Tensor):
        if self._action_rows_dev is None:
          self._action_rows_dev = t.from_numpy(self._action_rows).to(self._eng.device)
        # (no host synchronisation on the device path: ids outside the table are the
        # caller's bug and are clamped to its ends; host arrays are validated below)
        self._eng.step_fields(self._action_rows_dev[a.long().clamp_(0, K - 1)].contiguous())
      else:
        a = a.astype(np.int64)
        if a.shape != (self._eng.N, self._eng.P):
          raise ValueError(f"actions must have shape {(self._eng.N, self._eng.P)}")
        if ((a < 0) | (a >= K)).any():
          raise ValueError(f"actions must be in [0, {K})")
        self._eng.step_fields(self._action_rows[a])
    self._submissions += 1
    return self._emit(self._timestep())

  def observables(self) -> SubstrateObservables:
    """substrate.py:102-104.  `events` emits (name, payload) like the reference —
    of world 0 when the substrate is batched; `events_batched` (batched substrates)
    emits (world, (name, payload)) for every world."""
    return self._observables

  def _emit(self, timestep: TimeStep) -> TimeStep:
    self._observables.timestep.on_next(timestep)
    batched = self._observables.events_batched
    if batched is not None and batched._observers:   # decoding costs a device read
      for world, events in enumerate(self._eng.events_all()):
        for event in events:
          batched.on_next((world, event))
    if self._observables.events._observers:
      for event in self.events(0):
        self._observables.events.on_next(event)
    return timestep

  def events(self, world: int = 0):
    """`Substrate.events()` (wrappers/base.py:72-74) of one world of the batch for
    the last reset()/step(): [(name, {key: int}), ...], canonical order.  The raw
    device tensor for all worlds is `engine.observe(engine.OBS_EVENTS)`."""
    return self._eng.events(world)

  def observation_spec(self) -> List[Mapping[str, Array]]:
    spec = dict(self._config.timestep_spec)
    spec["COLLECTIVE_REWARD"] = Array((), np.float64, "COLLECTIVE_REWARD")
    return [dict(spec) for _ in self._roles]

  def action_spec(self) -> List[DiscreteArray]:
    spec = self._config.action_spec
    if self._action_rows is not None:
      spec = DiscreteArray(len(self._action_rows), spec.dtype, spec.name)
    # (every player's is named 'action': discrete_action_wrapper.py:103-109)
    return [spec.replace(name="action") for _ in self._roles]

  def reward_spec(self) -> List[Array]:
    return [Array((), np.float64, f"{i + 1}.REWARD") for i in range(len(self._roles))]

  def discount_spec(self) -> BoundedArray:
    return BoundedArray((), np.float64, 0.0, 1.0, "discount")

  def close(self):
    if not self._closed:
      self._closed = True
      self._eng.close()
      for subject in (self._observables.action, self._observables.timestep,
                      self._observables.events, self._observables.events_batched):
        if subject is not None:
          subject.on_completed()

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()

  # -- shaping -------------------------------------------------------------
  def _timestep(self) -> TimeStep:
    cfg = self._config
    if self._batched and self._T:
      s = self.slot if self._submissions else 0
      ts = RolloutTimeStep(self._step_type[s], self._reward[s], self._discount[s],
                           {n: v[s] for n, v in self._obs.items()})
      ts.slot = s
      return ts
    if self._batched:
      obs = dict(self._obs)
      return TimeStep(self._step_type, self._reward, self._discount, obs)
    # one world: the reference's per-player list of dicts, numpy leaves
    t = self._eng._torch
    self._host.copy_(self._blob, non_blocking=True)
    if self._blob.is_cuda:
      t.cuda.current_stream(self._eng.device).synchronize()
    host = {k: self._host_views[k][0].copy() for k in self._obs}
    reward = self._host_views["#reward"][0].copy()
    per_player = []
    for p in range(self._eng.P):
      d = {}
      for n in cfg.individual_observation_names:
        d[n] = host[n][p]
      for n in cfg.global_observation_names:
        d[n] = host[n]
      d["COLLECTIVE_REWARD"] = host["COLLECTIVE_REWARD"]
      per_player.append(d)
    return TimeStep(StepType(int(self._host_views["#step_type"][0])),
                    [reward[p] for p in range(self._eng.P)],
                    float(self._host_views["#discount"][0]), per_player)


# --------------------------------------------------------------------------
# The factory surface (meltingpot/substrate.py:57-113,
# utils/substrates/substrate_factory.py:24-95, utils/substrates/substrate.py:107-139)

_EXTRA_SPECS = {"POSITION": Array((2,), np.int32, "POSITION"),
                "ORIENTATION": Array((), np.int32, "ORIENTATION"),
                # clean_up's debug metrics (clean_up.py:751-784)
                "PLAYER_CLEANED": Array((), np.float64, "PLAYER_CLEANED"),
                "PLAYER_ATE_APPLE": Array((), np.float64, "PLAYER_ATE_APPLE"),
                "NUM_OTHERS_PLAYER_ZAPPED_THIS_STEP": Array(
                    (), np.float64, "NUM_OTHERS_PLAYER_ZAPPED_THIS_STEP"),
                "NUM_OTHERS_WHO_ATE_THIS_STEP": Array((), np.float64, "NUM_OTHERS_WHO_ATE_THIS_STEP")}


def timestep_spec_of(observation_spec: Mapping[str, Array]) -> TimeStep:
  """utils/substrates/specs.py:149-166 `specs.timestep`: the spec of the timestep ONE
  player sees — step_type / reward / discount specs + the observation specs, each
  named after its key."""
  return TimeStep(
      step_type=BoundedArray((), np.int64, int(min(StepType)), int(max(StepType)), "step_type"),
      reward=Array((), np.float64, "reward"),
      discount=BoundedArray((), np.float64, 0, 1, "discount"),
      observation={n: sp.replace(name=n) for n, sp in observation_spec.items()})


def build_substrate(*, lab2d_settings: Mapping[str, Any],
                    individual_observations: Sequence[str],
                    global_observations: Sequence[str],
                    action_table: Sequence[Mapping[str, int]],
                    num_worlds: int = 1, **kwargs) -> Substrate:
  """utils/substrates/substrate.py:107-139 — on the HIP engine, for N worlds at once.

  `lab2d_settings` is ANY settings dict of a level the engine implements (what a
  reference config's `build(roles, config)` returns, or one the caller has edited): it
  is lowered here, at run time (`builder.lower_settings`), and the substrate runs THAT —
  no committed pack is consulted.  `individual_observations` / `global_observations`
  choose the leaves of a player's observation as in the reference (the multiplayer
  wrapper, multiplayer_wrapper.py:108-167; COLLECTIVE_REWARD is always added,
  collective_reward_wrapper.py:25-50); `action_table[i]` is what discrete action i does
  (discrete_action_wrapper.py:77-109; it is lowered into the pack, the lookup happens
  on the device).  The number of players is the settings' `numPlayers`.  Further
  keyword arguments are `Substrate`'s (num_worlds, env_seed, device, rollout_length...)."""
  from meltingpot_amd import builder as builder_lib   # (it imports this module)
  if not action_table:
    raise ValueError("action_table must not be empty")
  table = tuple(dict(row) for row in action_table)
  level, pack_bytes, config = builder_lib.lower_settings(lab2d_settings, action_set=table)
  from meltingpot_amd import pack as pack_lib
  tables = pack_lib.loads(pack_bytes)
  names = tuple(n.decode() for n in bytes(tables["action_names"]).split(b"\0")[:-1])
  ranges = tuple(tuple(int(v) for v in row) for row in tables["action_spec"].reshape(-1, 3))
  validate_action_table(table, names, ranges)   # discrete_action_wrapper.py:28-49
  individual = [n for n in config.individual_observation_names if n in set(individual_observations)]
  individual += [n for n in individual_observations if n not in individual]   # (unknown ones: refused below)
  spec = dict(config.timestep_spec)
  for n in list(individual) + list(global_observations):
    if n in _EXTRA_SPECS:
      spec[n] = _EXTRA_SPECS[n]
  config = SubstrateConfig(
      name=level, action_set=table, individual_observation_names=individual,
      global_observation_names=list(global_observations),
      timestep_spec={n: sp for n, sp in spec.items()
                     if n in set(individual) | set(global_observations)},
      valid_roles=config.valid_roles, default_player_roles=config.default_player_roles,
      aux0_name=config.aux0_name)
  return Substrate(config, config.default_player_roles, pack_bytes,
                   num_worlds=num_worlds, **kwargs)


class SubstrateFactory:
  """utils/substrates/substrate_factory.py:24-95, same constructor and methods;
  `build(roles, **kwargs)` also takes `Substrate`'s keyword arguments (`num_worlds`,
  `env_seed`, `rollout_length`...)."""

  def __init__(self, *, lab2d_settings_builder, individual_observations, global_observations,
               action_table, timestep_spec, action_spec, valid_roles, default_player_roles):
    self._lab2d_settings_builder = lab2d_settings_builder
    self._individual_observations = frozenset(individual_observations)
    self._global_observations = frozenset(global_observations)
    self._action_table = tuple(dict(row) for row in action_table)
    self._timestep_spec = timestep_spec
    self._action_spec = action_spec
    self._valid_roles = frozenset(valid_roles)
    self._default_player_roles = tuple(default_player_roles)
    self._packed = None   # (config, pack name): `from_packed_config`

  @classmethod
  def from_packed_config(cls, config: SubstrateConfig) -> "SubstrateFactory":
    """The factory of a substrate this package carries as a committed pack
    (`get_config(name)`, possibly edited): the config is checked against the pack —
    what it says and the pack cannot do is refused (`check_config_against_pack`), an
    edited `action_set` runs as a custom action table, edited observation lists pick
    the leaves."""
    # (like the reference configs' `timestep_spec`: without COLLECTIVE_REWARD, which the
    # built substrate's observation_spec() adds)
    obs = dict(config.timestep_spec)
    f = cls(lab2d_settings_builder=None,
            individual_observations=config.individual_observation_names,
            global_observations=config.global_observation_names,
            action_table=config.action_set, timestep_spec=timestep_spec_of(obs),
            action_spec=DiscreteArray(len(config.action_set)),
            valid_roles=config.valid_roles, default_player_roles=config.default_player_roles)
    f._packed = config
    return f

  def valid_roles(self):
    return self._valid_roles

  def default_player_roles(self):
    return self._default_player_roles

  def timestep_spec(self) -> TimeStep:
    return self._timestep_spec

  def action_spec(self) -> DiscreteArray:
    return self._action_spec

  def build(self, roles: Sequence[str], **kwargs) -> Substrate:
    if self._packed is not None:
      config = self._packed
      pack_bytes = engine_lib.load_pack(config.name)
      custom = check_config_against_pack(config, pack_bytes, len(roles))
      if custom is not None:
        kwargs.setdefault("action_table", custom)
      return Substrate(config, roles, pack_bytes, **kwargs)
    return build_substrate(
        lab2d_settings=self._lab2d_settings_builder(roles),
        individual_observations=self._individual_observations,
        global_observations=self._global_observations,
        action_table=self._action_table, **kwargs)


def check_config_against_pack(config: SubstrateConfig, pack_bytes: bytes, num_players: int):
  """What a (possibly edited) `SubstrateConfig` says, held against the committed pack
  it names.  Raises ValueError for what the pack cannot honour — observation specs of
  another geometry, more players than it was lowered for — so that an edited config
  never runs the stock substrate silently.  Returns the config's `action_set` when it
  differs from the pack's (it then runs as a custom action table), else None."""
  from meltingpot_amd import lower, pack as pack_lib
  t = pack_lib.loads(pack_bytes)
  hdr = t["hdr"]
  P = int(hdr[lower.HDR_P])
  if num_players > P:
    raise ValueError(f"{num_players} roles, but the committed pack of {config.name!r} was "
                     f"lowered for at most {P} players")
  S = int(hdr[lower.HDR_SPRITE])
  want = {"RGB": ((int(hdr[lower.HDR_VF]) + int(hdr[lower.HDR_VB]) + 1) * S,
                  (int(hdr[lower.HDR_VL]) + int(hdr[lower.HDR_VR]) + 1) * S, 3),
          "WORLD.RGB": (int(hdr[lower.HDR_H]) * S, int(hdr[lower.HDR_W]) * S, 3)}
  for n, shape in want.items():
    if n in config.timestep_spec and tuple(config.timestep_spec[n].shape) != shape:
      raise ValueError(
          f"config.timestep_spec[{n!r}] has shape {tuple(config.timestep_spec[n].shape)}, the "
          f"committed pack of {config.name!r} renders {shape}: a config that changes the map, "
          "the window or the sprite size needs its lab2d settings — build it with "
          "build_substrate(lab2d_settings=...) or from the reference's config "
          "(get_factory_from_config), which are lowered at run time")
  names = tuple(n.decode() for n in bytes(t["action_names"]).split(b"\0")[:-1])
  ranges = tuple(tuple(int(v) for v in row) for row in t["action_spec"].reshape(-1, 3))
  rows = validate_action_table(config.action_set, names, ranges)
  stock = np.asarray(t["action_table"], np.int32).reshape(-1, 4)[:, :len(names)]
  if rows.shape == stock.shape and np.array_equal(rows, stock):
    return None
  return tuple(dict(r) for r in config.action_set)


def get_factory_from_config(config) -> SubstrateFactory:
  """meltingpot/substrate.py:98-113.  `config` is either
    * a reference substrate config (an `ml_collections.ConfigDict` with
      `lab2d_settings_builder`, as `meltingpot.substrate.get_config` /
      `meltingpot.configs.substrates.get_config` return it — edited or not): the
      factory builds its lab2d settings for the roles and lowers THEM at run time,
      exactly what the config says; or
    * this package's `SubstrateConfig` (`get_config(name)`): the committed pack, with
      the config checked against it (`SubstrateFactory.from_packed_config`)."""
  if isinstance(config, SubstrateConfig):
    return SubstrateFactory.from_packed_config(config)
  if not hasattr(config, "lab2d_settings_builder"):
    raise TypeError("get_factory_from_config wants a SubstrateConfig of this package or a "
                    "reference substrate config with `lab2d_settings_builder`")

  def lab2d_settings_builder(roles):
    return config.lab2d_settings_builder(roles=roles, config=config)

  return SubstrateFactory(
      lab2d_settings_builder=lab2d_settings_builder,
      individual_observations=config.individual_observation_names,
      global_observations=config.global_observation_names,
      action_table=config.action_set,
      timestep_spec=config.timestep_spec,
      action_spec=config.action_spec,
      valid_roles=config.valid_roles,
      default_player_roles=config.default_player_roles)


def get_factory(name: str) -> SubstrateFactory:
  """meltingpot/substrate.py:92-95."""
  return get_factory_from_config(get_config(name))


def build(name: str, *, roles: Sequence[str], num_worlds: int = 1,
          **kwargs) -> Substrate:
  """reference: meltingpot/substrate.py:57-72 — plus `num_worlds` (and the other
  keyword arguments of `Substrate`)."""
  return get_factory(name).build(roles, num_worlds=num_worlds, **kwargs)


def build_from_config(config, *, roles: Sequence[str], num_worlds: int = 1,
                      **kwargs) -> Substrate:
  """reference: meltingpot/substrate.py:75-89.  The substrate that runs is the one the
  config DESCRIBES (`get_factory_from_config`): a reference config's own lab2d
  settings lowered at run time, or a `SubstrateConfig` checked against its pack."""
  return get_factory_from_config(config).build(roles, num_worlds=num_worlds, **kwargs)
