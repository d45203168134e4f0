"""Test infrastructure: the `engine.Engine` interface for ONE world, driven by
the CPU oracle.  It lets `lab2d_env.Environment` (product code) run where there
is no GPU, so that the reference's unmodified wrapper stack can be layered on it
in the CPU test suite (tests/test_reference_wrappers.py); the GPU suite then
checks that the HIP engine behind the same class gives the same timesteps."""
import numpy as np

from meltingpot_amd import engine as E
from oracle import oracle as oracle_lib


class OracleEngine:

  def __init__(self, pack_bytes: bytes, seed: int, num_players: int = 0):
    self._o = oracle_lib.Oracle(pack_bytes, seed, num_players)
    self.pack_bytes = pack_bytes
    self.P, self.N = self._o.P, 1
    self.num_actions = len(self._o.tables["action_table"]) // 4
    self._step_type = 0
    self._started = False

  def reset(self, seeds=None, mask=None):
    assert seeds is None and mask is None
    self._o.reset()
    self._started = True
    self._step_type = 0

  def step(self, actions):
    a = np.asarray(actions, np.int32).reshape(1, self.P)
    if a.min() < 0 or a.max() >= self.num_actions:
      raise ValueError("action outside the ACTION_SET")   # mp_step_host
    if self._o.done:          # auto_reset: the step after LAST restarts the episode
      self._o.reset()
      self._step_type = 0
      return
    self._step_type = 1 if self._o.step(a[0]) else 2

  def step_fields(self, fields):
    a = np.asarray(fields, np.int32).reshape(self.P, -1)
    spec = self._o.tables["action_spec"].reshape(-1, 3)
    if (a < spec[:, 0]).any() or (a > spec[:, 1]).any():
      raise ValueError("action field outside its range")   # mp_step_fields_host
    if self._o.done:
      self._o.reset()
      self._step_type = 0
      return
    self._step_type = 1 if self._o.step_fields(a) else 2

  def observe_host(self, kind: int) -> np.ndarray:
    o = self._o
    first = self._step_type == 0
    if kind == E.OBS_RGB:
      return np.stack([o.render_agent(p) for p in range(self.P)])[None]
    if kind == E.OBS_WORLD_RGB:
      return o.render_world()[None]
    if kind == E.OBS_REWARD:
      return (np.zeros(self.P) if first else o.rewards())[None]
    if kind == E.OBS_READY_TO_SHOOT:
      return o.ready_to_shoot()[None]
    if kind == E.OBS_AUX0:
      return o.num_others_cleaned()[None]
    if kind == E.OBS_INVENTORY:
      return o.inventories()[0][None]
    if kind == E.OBS_INTERACTION_INVENTORIES:
      return o.inventories()[1][None]
    if kind == E.OBS_STEP_TYPE:
      return np.array([self._step_type], np.int32)
    if kind == E.OBS_DISCOUNT:
      return np.array([1.0 if self._step_type == 1 else 0.0])
    if kind == E.OBS_COLLECTIVE_REWARD:
      return np.array([0.0 if first else o.rewards().sum()])
    raise KeyError(kind)

  def events(self, world: int = 0):
    assert world == 0
    out = []
    for t, a, b in self._o.events():
      name, keys = E.EVENT_TYPES[t]
      if t == 5 and b:
        keys = ("player_index", "class")
      payload = dict(zip(keys, (a, b)))
      if t == 11:   # the_matrix/components.lua:789-797
        rewards, inventories = self._o.interaction_rewards(), self._o.inventories()[1]
        payload.update(row_reward=float(rewards[a - 1, 0]), col_reward=float(rewards[a - 1, 1]),
                       row_inventory=inventories[a - 1, 0].copy(),
                       col_inventory=inventories[b - 1, 0].copy())
      out.append((name, payload))
    return out

  def close(self):
    self._o.close()
