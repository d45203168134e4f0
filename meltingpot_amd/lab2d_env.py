"""`dmlab2d.Environment` duck type over one world of the HIP engine.

The reference's wrapper stack (utils/substrates/substrate.py:129-139:
Observables -> Multiplayer -> DiscreteAction -> CollectiveReward -> Substrate)
talks to the object returned by `builder.builder()` through the flat
`"<lua player index>.<NAME>"` convention of dmlab2d
(utils/substrates/wrappers/base.py:38-84,
wrappers/multiplayer_wrapper.py:108-167).  This class offers that surface —
`reset / step / observation / events / action_spec / observation_spec /
reward_spec / discount_spec / list_property / read_property / write_property /
close` with `"N.move"`-style action dicts in and `"N.RGB"`, `"N.REWARD"`,
`"WORLD.RGB"`... observation dicts out — so that the UNMODIFIED reference
wrappers can be layered on the engine for conformance work (SURVEY.md §8f
row 1; tests/test_reference_wrappers.py does exactly that with the files under
/root/reference).  It is a compatibility path for one world on the host;
training loops should use `meltingpot_amd.substrate.build(..., num_worlds=N)`.
"""

from __future__ import annotations

from typing import Dict, Mapping

import numpy as np

from meltingpot_amd import engine as engine_lib
from meltingpot_amd import substrate as substrate_lib


class Environment:

  def __init__(self, name: str, roles, *, env_seed=None, device: int = 0, engine=None,
               config=None):
    """`engine`: an object with the `engine.Engine` interface for one world
    (default: a HIP engine on `device` with the committed pack of substrate `name`;
    the conformance tests pass a stand-in driven by the CPU oracle where there is no
    GPU).  `config`: the observation names and specs of a pack lowered at run time
    (`meltingpot_amd.builder`: `name` is then the level, `engine` runs that pack)."""
    if config is not None and engine is None:
      raise ValueError("a run-time config needs the engine created on its pack")
    self._cfg = config if config is not None else substrate_lib.get_config(name)
    invalid = set(roles) - self._cfg.valid_roles
    if invalid:
      raise ValueError(f"Invalid roles: {invalid!r}. Must be one of "
                       f"{self._cfg.valid_roles!r}")
    if not roles:
      raise ValueError("roles must not be empty")
    if engine is None:
      pack_bytes = engine_lib.load_pack(name)
      role_names = engine_lib.pack_role_names(pack_bytes)
      engine = engine_lib.Engine(
          pack_bytes, 1, device=device, auto_reset=True, num_players=len(roles),
          base_seed=substrate_lib.resolve_env_seed(env_seed), literal_seed=True,
          roles=[role_names.index(r) for r in roles] if role_names else None)
    self._eng = engine
    if self._eng.P != len(roles):
      raise ValueError(f"{len(roles)} roles for an engine of {self._eng.P} players")
    self._P = self._eng.P
    # actionOrder and actionSpec of the avatars (avatar_library.lua:205-223), as
    # lowered into the pack: the fields of dmlab2d's raw action surface
    self._names, self._ranges = substrate_lib.action_fields(self._eng)

  # -- dmlab2d.Environment surface -----------------------------------------
  def action_spec(self) -> Dict[str, substrate_lib.BoundedArray]:
    out = {}
    for p in range(self._P):
      for n, (lo, hi, _) in zip(self._names, self._ranges):
        out[f"{p + 1}.{n}"] = substrate_lib.BoundedArray(
            (), np.int32, lo, hi, f"{p + 1}.{n}")
    return out

  def observation_spec(self) -> Dict[str, substrate_lib.Array]:
    spec = {}
    for p in range(self._P):
      for n in self._cfg.individual_observation_names:
        spec[f"{p + 1}.{n}"] = self._cfg.timestep_spec[n].replace(name=f"{p + 1}.{n}")
      spec[f"{p + 1}.REWARD"] = substrate_lib.Array((), np.float64, f"{p + 1}.REWARD")
    for n in self._cfg.global_observation_names:
      spec[n] = self._cfg.timestep_spec[n]
    return spec

  def reward_spec(self) -> substrate_lib.Array:
    """dm_env.Environment.reward_spec default (dmlab2d.Environment inherits it): a
    float scalar; the per-player rewards travel as "N.REWARD" observations."""
    return substrate_lib.Array((), np.float64, "reward")

  def discount_spec(self) -> substrate_lib.BoundedArray:
    return substrate_lib.BoundedArray((), np.float64, 0.0, 1.0, "discount")

  # dmlab2d properties (wrappers/base.py:66-84): Melting Pot's levels register
  # none, so the tree is empty — the calls exist and answer like dmlab2d does for
  # an unknown key.
  def list_property(self, key: str = ""):
    if key:
      raise KeyError(key)
    return []

  def read_property(self, key: str):
    raise KeyError(key)

  def write_property(self, key: str, value):
    raise KeyError(key)

  def reset(self) -> substrate_lib.TimeStep:
    self._eng.reset()
    return self._timestep()

  def step(self, actions: Mapping[str, int]) -> substrate_lib.TimeStep:
    """dmlab2d.Environment.step: `actions` maps "<player>.<field>" to an int;
    any combination inside the action spec is an action (move + turn + fire in
    one step, level_playing_utils.py:283,333-334); a missing key keeps the
    field's default, an unknown key or a value outside its range is a
    ValueError (dmlab2d validates against its action spec)."""
    known = {f"{p + 1}.{n}" for p in range(self._P) for n in self._names}
    unknown = set(actions) - known
    if unknown:
      raise ValueError(f"unknown action keys {sorted(unknown)}")
    fields = np.zeros((1, self._P, len(self._names)), np.int32)
    for p in range(self._P):
      for a, (n, (lo, hi, default)) in enumerate(zip(self._names, self._ranges)):
        v = int(actions.get(f"{p + 1}.{n}", default))
        if not lo <= v <= hi:
          raise ValueError(f"{p + 1}.{n} = {v} is outside [{lo}, {hi}]")
        fields[0, p, a] = v
    self._eng.step_fields(fields)
    return self._timestep()

  def observation(self) -> Dict[str, np.ndarray]:
    E = engine_lib
    obs = {}
    per = {"RGB": E.OBS_RGB, "READY_TO_SHOOT": E.OBS_READY_TO_SHOOT,
           "INVENTORY": E.OBS_INVENTORY,
           "INTERACTION_INVENTORIES": E.OBS_INTERACTION_INVENTORIES}
    if self._cfg.aux0_name:
      per[self._cfg.aux0_name] = E.OBS_AUX0
    host = {n: self._eng.observe_host(per[n])[0]
            for n in self._cfg.individual_observation_names}
    reward = self._eng.observe_host(E.OBS_REWARD)[0]
    for p in range(self._P):
      for n in self._cfg.individual_observation_names:
        obs[f"{p + 1}.{n}"] = host[n][p]
      obs[f"{p + 1}.REWARD"] = reward[p]
    if "WORLD.RGB" in self._cfg.global_observation_names:
      obs["WORLD.RGB"] = self._eng.observe_host(E.OBS_WORLD_RGB)[0]
    return obs

  def events(self):
    """dmlab2d `env.events()`: [(name, [b'dict', b'key', value, ...]), ...]
    (the Lua side calls events:add(name, 'dict', key, value, ...))."""
    out = []
    for name, payload in self._eng.events(0):
      if not payload:
        out.append((name, [np.array(b"success")]))  # AvatarStarted: ('str', 'success')
        continue
      flat = [np.array(b"dict")]
      for k, v in payload.items():
        # (indices and classes are integers; the 'interaction' event's rewards and
        # inventories are doubles / DoubleTensors, the_matrix/components.lua:789-797)
        integral = isinstance(v, (int, np.integer))
        flat += [np.array(k.encode()), np.array(v, np.int64 if integral else np.float64)]
      out.append((name, flat))
    return out

  def close(self):
    self._eng.close()

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()

  # ------------------------------------------------------------------------
  def _timestep(self) -> substrate_lib.TimeStep:
    E = engine_lib
    st = substrate_lib.StepType(int(self._eng.observe_host(E.OBS_STEP_TYPE)[0]))
    # dmlab2d reports reward=None and discount=None on FIRST; the multiplayer
    # wrapper turns the None discount into 0.0 (multiplayer_wrapper.py:117)
    discount = None if st == substrate_lib.StepType.FIRST else float(
        self._eng.observe_host(E.OBS_DISCOUNT)[0])
    return substrate_lib.TimeStep(st, None, discount, self.observation())
