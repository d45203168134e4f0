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
    # the product's own decoder (engine.Engine._decode_events) over the oracle's rows
    ev = list(self._o.events())
    rows = np.zeros((1 + len(ev), 4), np.int64)
    rows[0, 0] = len(ev)
    for i, (t, a, b) in enumerate(ev):
      rows[1 + i, :3] = (t, a, b)
    interaction = None
    if any(t == 11 for t, _, _ in ev):   # the_matrix/components.lua:789-797
      interaction = (self._o.interaction_rewards(), self._o.inventories()[1])
    return E.Engine._decode_events(rows, 0, interaction, E.pack_agent_roles(self.pack_bytes))

  def close(self):
    self._o.close()


class OracleBatchEngine:
  """Test infrastructure: the WHOLE `engine.Engine` interface `substrate.Substrate` uses
  — N worlds, bound torch tensors (on the CPU), rings — driven by N CPU oracles.  The CPU
  suite monkeypatches `meltingpot_amd.substrate.engine_lib.Engine` with this class so that
  the product's `Substrate` / `SubstrateFactory` / `build_substrate` run where there is no
  GPU (and the reference's `Scenario` on top of them); the GPU suite runs the same code on
  the HIP engine and compares the timesteps leaf by leaf."""

  PLACE_MIN_BYTES = 1 << 62

  def __init__(self, pack_bytes, num_worlds, *, device=0, auto_reset=True, world_offset=0,
               base_seed=0, literal_seed=False, num_players=0, debug_observations=False,
               unfused=None, dev=None, roles=None, placements=24):
    import types
    import torch
    from meltingpot_amd import lower, pack as pack_lib
    assert literal_seed or base_seed, "the stand-in takes the Substrate API's seeds"
    tables = pack_lib.loads(pack_bytes)
    if roles is not None:
      pack_bytes = pack_lib.dumps(lower.apply_roles(tables, list(roles)))
      num_players = len(roles)
    self._torch = torch
    self.device = torch.device("cpu")
    self.pack_bytes = pack_bytes
    self.N = int(num_worlds)
    self._auto_reset = auto_reset
    self._o = [oracle_lib.Oracle(pack_bytes, (int(base_seed) + world_offset + w) % (1 << 64), num_players)
               for w in range(self.N)]
    o = self._o[0]
    self.P = o.P
    t = o.tables
    hdr = t["hdr"]
    self.num_actions = len(t["action_table"]) // 4
    S = int(hdr[lower.HDR_SPRITE])
    vh = int(hdr[lower.HDR_VF]) + int(hdr[lower.HDR_VB]) + 1
    vw = int(hdr[lower.HDR_VL]) + int(hdr[lower.HDR_VR]) + 1
    H, W = int(hdr[lower.HDR_H]), int(hdr[lower.HDR_W])
    R = ((len(t["mx_states"]) - 8) // 2 if "mx_states" in t
         else int(t["gr_i32"][7]) if "gr_i32" in t else 0)
    self.info = types.SimpleNamespace(num_action_fields=int(hdr[lower.HDR_NFIELDS]),
                                      num_resources=R, num_worlds=self.N, num_players=self.P)
    N, P = self.N, self.P
    self.shapes = {
        E.OBS_RGB: ((N, P, vh * S, vw * S, 3), torch.uint8),
        E.OBS_WORLD_RGB: ((N, H * S, W * S, 3), torch.uint8),
        E.OBS_REWARD: ((N, P), torch.float64),
        E.OBS_READY_TO_SHOOT: ((N, P), torch.float64),
        E.OBS_AUX0: ((N, P), torch.float64),
        E.OBS_STEP_TYPE: ((N,), torch.int32),
        E.OBS_DISCOUNT: ((N,), torch.float64),
        E.OBS_COLLECTIVE_REWARD: ((N,), torch.float64),
        E.OBS_POSITION: ((N, P, 2), torch.int32),
        E.OBS_ORIENTATION: ((N, P), torch.int32),
        E.OBS_INVENTORY: ((N, P, R), torch.float64),
        E.OBS_INTERACTION_INVENTORIES: ((N, P, 2, R), torch.float64),
    }
    self._bound = {}
    self._ring = {}
    self._cursor = 0
    self._step_type = np.zeros(N, np.int32)
    self.placement = {}

  # -- buffers
  def empty(self, kind):
    shape, dtype = self.shapes[kind]
    return self._torch.zeros(shape, dtype=dtype)

  def bind(self, kind, tensor=None):
    if tensor is None:
      tensor = self.empty(kind)
    assert tuple(tensor.shape) == self.shapes[kind][0] and tensor.dtype == self.shapes[kind][1]
    self._bound[kind] = tensor
    self._ring.pop(kind, None)
    return tensor

  def bind_ring(self, kind, tensor=None, slots=None, tune=True):
    shape, dtype = self.shapes[kind]
    if tensor is None:
      tensor = self._torch.zeros((int(slots),) + tuple(shape), dtype=dtype)
    if self._ring and tensor.shape[0] != next(iter(self._ring.values())).shape[0]:
      raise ValueError("one slot count for all ring kinds")
    if not self._ring:
      self._cursor = 0
    self._ring[kind] = tensor
    self._bound.pop(kind, None)
    return tensor

  @property
  def ring(self):
    T = next(iter(self._ring.values())).shape[0] if self._ring else 0
    nxt = self._cursor % T if T else 0
    return {"slots": T, "next": nxt, "last": (nxt + T - 1) % T if T else 0}

  def use_current_stream(self):
    pass

  def close(self):
    for o in self._o:
      o.close()
    self._o = []

  # -- stepping
  def reset(self, seeds=None, mask=None):
    assert seeds is None and mask is None
    for w, o in enumerate(self._o):
      o.reset()
      self._step_type[w] = 0
    self._publish()

  def _advance(self, w, fn):
    o = self._o[w]
    if o.done:          # the step after LAST restarts the episode (auto_reset)
      assert self._auto_reset
      o.reset()
      self._step_type[w] = 0
    else:
      self._step_type[w] = 1 if fn(o) else 2

  def step(self, actions):
    a = np.asarray(actions.cpu() if hasattr(actions, "cpu") else actions, np.int32).reshape(self.N, self.P)
    if a.min() < 0 or a.max() >= self.num_actions:
      raise ValueError("action outside the ACTION_SET")
    for w in range(self.N):
      self._advance(w, lambda o: o.step(a[w]))
    self._publish()

  def step_fields(self, fields):
    f = np.asarray(fields.cpu() if hasattr(fields, "cpu") else fields, np.int32).reshape(self.N, self.P, -1)
    for w in range(self.N):
      self._advance(w, lambda o: o.step_fields(f[w]))
    self._publish()

  def _value(self, kind):
    first = self._step_type == 0
    per = lambda fn: np.stack([fn(o) for o in self._o])
    if kind == E.OBS_RGB:
      return per(lambda o: np.stack([o.render_agent(p) for p in range(self.P)]))
    if kind == E.OBS_WORLD_RGB:
      return per(lambda o: o.render_world())
    if kind == E.OBS_REWARD:
      return np.where(first[:, None], 0.0, per(lambda o: o.rewards()))
    if kind == E.OBS_READY_TO_SHOOT:
      return per(lambda o: o.ready_to_shoot())
    if kind == E.OBS_AUX0:
      return per(lambda o: o.num_others_cleaned())
    if kind == E.OBS_INVENTORY:
      return per(lambda o: o.inventories()[0])
    if kind == E.OBS_INTERACTION_INVENTORIES:
      return per(lambda o: o.inventories()[1])
    if kind == E.OBS_STEP_TYPE:
      return self._step_type.copy()
    if kind == E.OBS_DISCOUNT:
      return (self._step_type == 1).astype(np.float64)
    if kind == E.OBS_COLLECTIVE_REWARD:
      return np.where(first, 0.0, per(lambda o: o.rewards()).sum(axis=1))
    if kind == E.OBS_POSITION:
      return per(lambda o: o.dump()[1][:, :2]).astype(np.int32)
    if kind == E.OBS_ORIENTATION:
      return per(lambda o: o.dump()[1][:, 2]).astype(np.int32)
    raise KeyError(kind)

  def _publish(self):
    t = self._torch
    for kind, tensor in self._bound.items():
      tensor.copy_(t.from_numpy(np.ascontiguousarray(self._value(kind))).view(tensor.shape))
    if self._ring:
      T = next(iter(self._ring.values())).shape[0]
      s = self._cursor % T
      for kind, tensor in self._ring.items():
        tensor[s].copy_(t.from_numpy(np.ascontiguousarray(self._value(kind))).view(tensor[s].shape))
      self._cursor += 1

  def observe(self, kind, out=None):
    v = self._torch.from_numpy(np.ascontiguousarray(self._value(kind)))
    if out is None:
      return v
    out.copy_(v.view(out.shape))
    return out

  def observe_host(self, kind):
    return np.asarray(self._value(kind))

  def events(self, world=0):
    single = OracleEngine.__new__(OracleEngine)
    single._o = self._o[world]
    return OracleEngine.events(single, 0)

  def events_all(self, worlds=None):
    return [self.events(w) for w in (range(self.N) if worlds is None else worlds)]
