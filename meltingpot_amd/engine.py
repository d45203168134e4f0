"""ctypes binding of libmp_engine.so (C ABI: include/mp_engine.h).

Thin by design: the engine is the product, this module only moves pointers.
PyTorch-ROCm supplies device memory for action / observation tensors and the
stream; nothing here computes or measures (where a bound view is allocated and
which launch plan suits it are the library's business: mp_place_output, mp_tune).
There is no CPU fallback — constructing an `Engine` without a GPU (or without the
built library) raises.

Reference boundary replaced: `dmlab2d.Lab2d(...)` / `dmlab2d.Environment(...)`
(meltingpot/utils/substrates/builder.py:179-187).
"""

from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional, Sequence

import numpy as np

from meltingpot_amd import _build

OBS_RGB = 0
OBS_WORLD_RGB = 1
OBS_REWARD = 2
OBS_READY_TO_SHOOT = 3
OBS_AUX0 = 4
OBS_STEP_TYPE = 5
OBS_DISCOUNT = 6
OBS_COLLECTIVE_REWARD = 7
OBS_POSITION = 8
OBS_ORIENTATION = 9
OBS_EVENTS = 10
# debug observations (produced while bound, or with debug_observations=True)
OBS_AUX1 = 11  # clean_up: PLAYER_CLEANED
OBS_AUX2 = 12  # clean_up: PLAYER_ATE_APPLE
OBS_AUX3 = 13  # clean_up: NUM_OTHERS_PLAYER_ZAPPED_THIS_STEP
OBS_AUX4 = 14  # clean_up: NUM_OTHERS_WHO_ATE_THIS_STEP
OBS_ZAP_MATRIX = 15
OBS_LAYER = 16
OBS_INVENTORY = 17                 # *_in_the_matrix: "N.INVENTORY" f64 [N, P, R]
OBS_INTERACTION_INVENTORIES = 18   # "N.INTERACTION_INVENTORIES" f64 [N, P, 2, R]
# *_in_the_matrix debug cumulants f64 [N, P, 1 + 3 R] (the_matrix.py:22-60); columns:
OBS_MATRIX_CUMULANTS = 19
# *_in_the_matrix: f64 [N, P, 2], (row_reward, col_reward) of the latest interaction of
# player p — the rest of the 'interaction' event's payload (include/mp_engine.h)
OBS_INTERACTION_REWARDS = 20


def matrix_cumulant_names(num_resources: int):
  """Reference observation names of the columns of OBS_MATRIX_CUMULANTS."""
  names = ["INTERACTED_THIS_STEP"]
  for k in range(1, num_resources + 1):
    names += [f"COLLECTED_RESOURCE_{k}", f"DESTROYED_RESOURCE_{k}",
              f"ARGMAX_INTERACTION_INVENTORY_WAS_{k}"]
  return names
EVENT_ROWS = 128  # MP_EVENT_ROWS: 1 header row + up to 127 events per world-step
# MpEventType -> (reference event name, payload keys)  (include/mp_engine.h)
EVENT_TYPES = {
    1: ("zap", ("source", "target")),
    2: ("edible_consumed", ("player_index",)),
    3: ("player_cleaned", ("player_index",)),
    4: ("claimed_resource", ("player_index",)),
    5: ("destroyed_resource", ("player_index",)),
    6: ("sanctioning", ("source", "target")),
    7: ("removal_due_to_sanctioning", ("source", "target")),
    8: ("set_sanctioning_level", ("player_index", "level")),
    9: ("AvatarStarted", ()),
    # payload b = player_coin_type << 1 | coin_type, indices of the two coin colours
    10: ("coin_consumed", ("player_index", "types")),
    # the_matrix: + row_reward, col_reward, row_inventory, col_inventory
    # (the_matrix/components.lua:789-797), read from OBS_INTERACTION_REWARDS and
    # OBS_INTERACTION_INVENTORIES of the same step by `Engine.events`
    11: ("interaction", ("row_player_idx", "col_player_idx")),
    12: ("collected_resource", ("player_index", "class")),
    # coop_mining/components.lua:196,210,220 (ore_type: 1 iron, 2 gold)
    13: ("mining", ("player", "ore_type")),
    14: ("extraction", ("player", "ore_type")),
    # payload b = player_b << 2 | ore_type (decoded by `Engine.events`)
    15: ("extraction_pair", ("player_a", "player_b", "ore_type")),
    # gift_refinements/components.lua:174-181; a = gifter_index | source_type << 4,
    # b = receipient_index | received_amount << 4 (decoded by `Engine.events`, which adds the
    # two avatars' roles from the pack's "agent_roles"; the reference's spelling of
    # "receipient" is kept)
    16: ("gift", ("gifter_index", "gifter_role", "receipient_index", "receipient_role",
                  "source_type", "received_amount")),
    # collaborative_cooking/components.lua:325-328, 397-400, 412-415 (item: 1 tomato, 2 dish,
    # 3 soup; decoded to the reference's strings by `Engine.events`)
    # externality_mushrooms/components.lua:72-74 (the type decoded to its state's name by
    # `Engine.events`)
    20: ("eating_mushroom", ("player_index", "mushroom_type")),
    17: ("receiver_accepted_item", ("player_index", "item")),
    18: ("item_dropped_into_pot", ("player_index", "item")),
    19: ("cooked_food_collected_from_pot", ("player_index", "cooked_item")),
}
COOKING_ITEMS = ("empty", "tomato", "dish", "soup")
# the live states of externality_mushrooms' mushroom prefab (externality_mushrooms.py:520-545)
MUSHROOM_TYPES = ("fullInternalityZeroExternality", "halfInternalityHalfExternality",
                  "zeroInternalityFullExternality", "negativeInternalityNegativeExternality")

COUNTER_NAMES = ("world_steps", "agent_steps", "episodes", "reward_sum_x1024",
                 "zaps", "aux0", "respawns", "bad_actions")

MP_ERR_INVALID = -1
MP_ERR_NO_DEVICE = -3
MP_ABI_VERSION = 8

# Every symbol include/mp_engine.h declares (tests check the library exports
# exactly these).
ABI_SYMBOLS = (
    "mp_abi_version", "mp_last_error", "mp_create", "mp_destroy", "mp_info",
    "mp_set_stream", "mp_bind_output", "mp_reset", "mp_step", "mp_step_host",
    "mp_step_fields", "mp_step_fields_host",
    "mp_observe", "mp_obs_bytes", "mp_dump", "mp_snapshot_bytes",
    "mp_snapshot", "mp_restore", "mp_counters", "mp_sync", "mp_fault_words",
    "mp_alloc_output", "mp_free_output", "mp_tune", "mp_place_output",
    "mp_bind_output_ring", "mp_set_retired_va_limit",
    "mp_torch_alloc", "mp_torch_free", "mp_box_fill")


class MpDevOptions(ctypes.Structure):
  """Test / development overrides of the launch plan (include/mp_engine.h);
  product code never passes one."""
  _fields_ = [("struct_size", ctypes.c_uint32)] + [(n, ctypes.c_int32) for n in (
      "batch_worlds", "waves", "feeders", "max_groups", "scratch_cells",
      "no_composite_cache", "max_composites", "verbose", "late_feeder_prio",
      "ring_batches", "static_pct", "world_waves", "store_sc1", "head", "no_next_orders",
      "record_pad", "pace", "team")]


class MpConfig(ctypes.Structure):
  _fields_ = [
      ("struct_size", ctypes.c_uint32),
      ("device", ctypes.c_int32),
      ("num_worlds", ctypes.c_int32),
      ("auto_reset", ctypes.c_int32),
      ("world_offset", ctypes.c_uint64),
      ("base_seed", ctypes.c_uint64),
      ("stream", ctypes.c_void_p),
      ("num_players", ctypes.c_int32),
      ("debug_observations", ctypes.c_int32),
      ("unfused", ctypes.c_int32),
      ("literal_base_seed", ctypes.c_int32),
      ("dev", ctypes.POINTER(MpDevOptions)),
      ("roles", ctypes.POINTER(ctypes.c_int32)),
  ]


class MpInfo(ctypes.Structure):
  _fields_ = [(n, ctypes.c_int32) for n in (
      "abi_version", "substrate", "num_worlds", "num_players", "num_actions",
      "map_h", "map_w", "num_layers", "sprite_size", "view_h", "view_w",
      "max_frames", "world_state_bytes", "fused", "num_resources",
      "num_action_fields", "plan_batch_worlds", "plan_ring_batches", "plan_owned_batches",
      "plan_pooled_batches", "plan_groups", "plan_store_sc1", "plan_feeders", "plan_waves",
      "ring_slots", "ring_next", "plan_pace", "visible_layers", "plan_team", "plan_late_priority")] + [("retired_va_bytes", ctypes.c_int64),
                                      ("retired_va_limit", ctypes.c_int64)]


class MpPlacement(ctypes.Structure):
  _fields_ = [("candidates", ctypes.c_int32), ("picked", ctypes.c_int32),
              ("us", ctypes.c_float * 32), ("stepped", ctypes.c_int32),
              ("requested", ctypes.c_int32), ("out_of_memory", ctypes.c_int32),
              ("early_exit", ctypes.c_int32), ("setup_ms", ctypes.c_float)]


class MpBoxFill(ctypes.Structure):
  _fields_ = [("bytes", ctypes.c_uint64), ("memset_us", ctypes.c_float),
              ("product_order_us", ctypes.c_float), ("front_4k_us", ctypes.c_float),
              ("groups", ctypes.c_int32), ("waves", ctypes.c_int32), ("span_bytes", ctypes.c_uint32)]


class EngineError(RuntimeError):
  pass


_lib = None


def load_library(build: bool = True) -> ctypes.CDLL:
  """Loads libmp_engine.so (building it with hipcc first if needed)."""
  global _lib
  if _lib is not None:
    return _lib
  # torch first: its wheel bundles its own HIP runtime, and a process must end up
  # with ONE — loaded the other way round (this library pulling in /opt/rocm's
  # runtime, torch its own afterwards) the second runtime finds no device
  try:
    import torch  # noqa: F401
  except ImportError:
    pass
  path = _build.LIB_PATH
  override = os.environ.get("MP_ENGINE_LIB")  # developer A/B runs of another build
  if override:
    path = override
  elif build:
    path = _build.build_engine()
  if not os.path.exists(path):
    raise EngineError(
        f"{path} is missing: build it with `python -c 'import __graft_entry__ "
        "as g; g.build()'` (hipcc --offload-arch=gfx950). There is no fallback.")
  L = ctypes.CDLL(path)
  vp, i32, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64
  L.mp_abi_version.restype = i32
  L.mp_last_error.restype = ctypes.c_char_p
  L.mp_create.restype = i32
  L.mp_create.argtypes = [vp, u64, ctypes.POINTER(MpConfig),
                          ctypes.POINTER(vp)]
  L.mp_destroy.restype = None
  L.mp_destroy.argtypes = [vp]
  L.mp_info.restype = i32
  L.mp_info.argtypes = [vp, ctypes.POINTER(MpInfo)]
  L.mp_set_stream.restype = i32
  L.mp_set_stream.argtypes = [vp, vp]
  L.mp_bind_output.restype = i32
  L.mp_bind_output.argtypes = [vp, i32, vp]
  L.mp_reset.restype = i32
  L.mp_reset.argtypes = [vp, vp, vp]
  L.mp_step.restype = i32
  L.mp_step.argtypes = [vp, vp]
  L.mp_step_host.restype = i32
  L.mp_step_host.argtypes = [vp, vp]
  L.mp_step_fields.restype = i32
  L.mp_step_fields.argtypes = [vp, vp]
  L.mp_step_fields_host.restype = i32
  L.mp_step_fields_host.argtypes = [vp, vp]
  L.mp_observe.restype = i32
  L.mp_observe.argtypes = [vp, i32, vp]
  L.mp_obs_bytes.restype = u64
  L.mp_obs_bytes.argtypes = [vp, i32]
  L.mp_dump.restype = i32
  L.mp_dump.argtypes = [vp, vp, vp, vp]
  L.mp_snapshot_bytes.restype = u64
  L.mp_snapshot_bytes.argtypes = [vp]
  L.mp_snapshot.restype = i32
  L.mp_snapshot.argtypes = [vp, vp, u64]
  L.mp_restore.restype = i32
  L.mp_restore.argtypes = [vp, vp, u64]
  L.mp_counters.restype = i32
  L.mp_counters.argtypes = [vp, vp]
  L.mp_sync.restype = i32
  L.mp_sync.argtypes = [vp]
  L.mp_alloc_output.restype = i32
  L.mp_alloc_output.argtypes = [i32, u64, u64, ctypes.POINTER(vp)]
  L.mp_free_output.restype = i32
  L.mp_free_output.argtypes = [i32, vp]
  L.mp_tune.restype = i32
  L.mp_tune.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
  L.mp_place_output.restype = i32
  L.mp_place_output.argtypes = [vp, i32, i32, u64, ctypes.POINTER(vp), ctypes.POINTER(MpPlacement)]
  L.mp_fault_words.restype = i32
  L.mp_fault_words.argtypes = [vp, vp]
  L.mp_bind_output_ring.restype = i32
  L.mp_bind_output_ring.argtypes = [vp, i32, vp, u64, i32]
  L.mp_box_fill.restype = i32
  L.mp_box_fill.argtypes = [vp, i32, i32, ctypes.POINTER(MpBoxFill)]
  L.mp_set_retired_va_limit.restype = i32
  L.mp_set_retired_va_limit.argtypes = [ctypes.c_int64]
  _lib = L
  return L


def _check(L, rc: int, what: str):
  if rc == 0:
    return
  msg = (L.mp_last_error() or b"").decode()
  if rc == MP_ERR_INVALID:
    raise ValueError(f"{what}: {msg}")  # the reference raises ValueError too
  raise EngineError(f"{what} failed ({rc}): {msg}")


class Engine:
  """N worlds of one substrate on one GPU.  All buffers are torch tensors on
  that GPU; calls enqueue work on torch's current stream and do not sync."""

  def __init__(self, pack_bytes: bytes, num_worlds: int, *, device: int = 0,
               auto_reset: bool = True, world_offset: int = 0,
               base_seed: int = 0, literal_seed: bool = False, num_players: int = 0,
               debug_observations: bool = False, unfused: Optional[bool] = None,
               dev: Optional[Dict[str, int]] = None,
               roles: Optional[Sequence[int]] = None, placements: int = 8):
    """`num_players` = 0: the pack's default count (its header; all the avatars
    it holds unless tools/make_packs.py says otherwise); else the first
    `num_players` avatars play (the reference's num_players = len(roles)).
    `base_seed`: world w is seeded base_seed + w (mod 2**64); base_seed 0 selects
    the benchmark's fixed per-world seeds unless `literal_seed` says that 0 is a
    seed like any other (MpConfig.literal_base_seed: the Substrate API's env_seed).
    `unfused`: True = one launch for the rules and one per view, False = one
    fused launch per step, None = the engine's choice for the substrate
    (`info.fused` reports it).  `dev`: MpDevOptions fields by name — tests and
    tools/ only (launch-plan overrides; results never depend on them).  `roles`:
    one index per player into the pack's "role_names" (MpConfig.roles; only for
    substrates whose config has more than one valid role) — see
    `pack_role_names`."""
    import torch  # device memory + streams only
    self._torch = torch
    # candidates `place` tries for a bound pixel view (1: the first allocation, its
    # plan tuned; 0: the first allocation, stock plan)
    self.placements = placements
    # memory `place` may keep alive while it probes (0: a quarter of the device's free
    # memory) — one process per GPU sets nothing; ranks that SHARE a device set their share
    self.place_max_bytes = 0
    self.placement: Dict[int, dict] = {}
    self._L = load_library()
    if not torch.cuda.is_available():
      # still go through mp_create so that the C ABI reports the error
      pass
    self._pack = ctypes.create_string_buffer(pack_bytes, len(pack_bytes))
    self.pack_bytes = pack_bytes
    stream = None
    if torch.cuda.is_available():
      torch.cuda.set_device(device)
      stream = torch.cuda.current_stream(device).cuda_stream
    cfg = MpConfig(ctypes.sizeof(MpConfig), device, num_worlds,
                   1 if auto_reset else 0, world_offset, int(base_seed) % (1 << 64), stream,
                   int(num_players), 1 if debug_observations else 0,
                   0 if unfused is None else (1 if unfused else 2),
                   1 if literal_seed else 0, None, None)
    if roles is not None:
      if num_players and len(roles) != num_players:
        raise ValueError(f"{len(roles)} roles for {num_players} players")
      self._roles = (ctypes.c_int32 * len(roles))(*[int(r) for r in roles])
      cfg.roles = ctypes.cast(self._roles, ctypes.POINTER(ctypes.c_int32))
      cfg.num_players = len(roles)
    if dev:
      opts = MpDevOptions(ctypes.sizeof(MpDevOptions), max_composites=-1)
      for k, v in dev.items():
        if not hasattr(opts, k) or k == "struct_size":
          raise ValueError(f"unknown MpDevOptions field {k!r}")
        setattr(opts, k, int(v))
      self._dev = opts   # kept alive for mp_create
      cfg.dev = ctypes.pointer(opts)
    handle = ctypes.c_void_p()
    rc = self._L.mp_create(self._pack, len(pack_bytes), ctypes.byref(cfg),
                           ctypes.byref(handle))
    self._h = None
    _check(self._L, rc, "mp_create")
    self._h = handle
    info = MpInfo()
    _check(self._L, self._L.mp_info(self._h, ctypes.byref(info)), "mp_info")
    self.info = info
    self.device = torch.device("cuda", device)
    self.N, self.P = info.num_worlds, info.num_players
    self.num_actions = info.num_actions
    S = info.sprite_size
    self.shapes = {
        OBS_RGB: ((self.N, self.P, info.view_h * S, info.view_w * S, 3), torch.uint8),
        OBS_WORLD_RGB: ((self.N, info.map_h * S, info.map_w * S, 3), torch.uint8),
        OBS_REWARD: ((self.N, self.P), torch.float64),
        OBS_READY_TO_SHOOT: ((self.N, self.P), torch.float64),
        OBS_AUX0: ((self.N, self.P), torch.float64),
        OBS_STEP_TYPE: ((self.N,), torch.int32),
        OBS_DISCOUNT: ((self.N,), torch.float64),
        OBS_COLLECTIVE_REWARD: ((self.N,), torch.float64),
        OBS_POSITION: ((self.N, self.P, 2), torch.int32),
        OBS_ORIENTATION: ((self.N, self.P), torch.int32),
        OBS_EVENTS: ((self.N, EVENT_ROWS, 4), torch.int32),
        OBS_AUX1: ((self.N, self.P), torch.float64),
        OBS_AUX2: ((self.N, self.P), torch.float64),
        OBS_AUX3: ((self.N, self.P), torch.float64),
        OBS_AUX4: ((self.N, self.P), torch.float64),
        OBS_ZAP_MATRIX: ((self.N, self.P, self.P), torch.float64),
        OBS_LAYER: ((self.N, self.P, info.view_h, info.view_w, info.num_layers),
                    torch.int32),
        OBS_INVENTORY: ((self.N, self.P, info.num_resources), torch.float64),
        OBS_INTERACTION_INVENTORIES: ((self.N, self.P, 2, info.num_resources), torch.float64),
        OBS_MATRIX_CUMULANTS: ((self.N, self.P, 1 + 3 * info.num_resources), torch.float64),
        OBS_INTERACTION_REWARDS: ((self.N, self.P, 2), torch.float64),
    }
    self._bound: Dict[int, "torch.Tensor"] = {}

  @property
  def fused(self) -> bool:
    """Whether a step with the views bound right now is ONE launch (rules and
    pixels fused) — MpConfig.unfused; the engine's own choice depends on the view."""
    info = MpInfo()
    _check(self._L, self._L.mp_info(self._h, ctypes.byref(info)), "mp_info")
    return bool(info.fused)

  @property
  def plan(self) -> Dict[str, int]:
    """The launch plan of a step with the pixel views bound right now (MpInfo.plan_*)."""
    info = MpInfo()
    _check(self._L, self._L.mp_info(self._h, ctypes.byref(info)), "mp_info")
    return {"batch_worlds": info.plan_batch_worlds, "ring_batches": info.plan_ring_batches,
            "owned_batches": info.plan_owned_batches, "pooled_batches": info.plan_pooled_batches,
            "workgroups": info.plan_groups, "sc1_stores": info.plan_store_sc1,
            "feeders": info.plan_feeders, "waves": info.plan_waves, "pace": info.plan_pace,
            "xcd_teams": info.plan_team, "late_feeder_priority": info.plan_late_priority}

  # -- lifetime ------------------------------------------------------------
  def close(self):
    if getattr(self, "_h", None):
      self._L.mp_destroy(self._h)
      self._h = None
      self._bound.clear()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()

  # -- events ------------------------------------------------------------------
  def events(self, world: int = 0):
    """env.events() of one world for the last reset()/step(): a list of
    (name, {key: value}) in canonical (sorted) order — the engine resolves a step's
    beams in parallel, so rows carry no order of their own.

    Payloads are the keys of the reference's events.  The_matrix's 'interaction'
    carries row_player_idx, col_player_idx, row_reward, col_reward (floats) and
    row_inventory, col_inventory (float64 arrays of R) as in the reference
    (the_matrix/components.lua:789-797)."""
    return self.events_all(worlds=[world])[0]

  def events_all(self, worlds=None):
    """events() of every world (or of `worlds`), from one device read per kind: a
    list of lists."""
    if worlds is None:
      worlds, pick = list(range(self.N)), (lambda t: t)
    else:
      worlds = [int(w) for w in worlds]
      index = self._torch.as_tensor(worlds, dtype=self._torch.long, device=self.device)
      pick = lambda t: t.index_select(0, index)   # (only these worlds cross to the host)
    rows = pick(self.observe(OBS_EVENTS)).cpu().numpy()
    extra = None
    if self.info.num_resources and any(
        (rows[i, 1:1 + int(rows[i, 0, 0]), 0] == 11).any() for i in range(len(worlds))):
      extra = (pick(self.observe(OBS_INTERACTION_REWARDS)).cpu().numpy(),
               pick(self.observe(OBS_INTERACTION_INVENTORIES)).cpu().numpy())
    roles = pack_agent_roles(self.pack_bytes) if (rows[:, 1:, 0] == 16).any() else None
    return [self._decode_events(rows[i], w, None if extra is None else (extra[0][i], extra[1][i]),
                                roles)
            for i, w in enumerate(worlds)]

  @staticmethod
  def _decode_events(rows, world, interaction=None, agent_roles=None):
    """`interaction`: this world's ([P, 2] rewards, [P, 2, R] inventories) when a row
    of type 11 is present; `agent_roles`: the avatars' agentRole strings (gift_refinements)."""
    n = int(rows[0, 0])
    if rows[0, 1]:
      raise EngineError(f"world {world}: {int(rows[0, 1])} events beyond the "
                        f"{EVENT_ROWS - 1} rows of MP_OBS_EVENTS were dropped")
    out = []
    for t, a, b, _ in sorted(tuple(int(v) for v in r) for r in rows[1:1 + n]):
      name, keys = EVENT_TYPES[t]
      if t == 5 and b:   # the_matrix's destroyed_resource names the class too (components.lua:178)
        keys = ("player_index", "class")
      payload = dict(zip(keys, (a, b)))
      if t == 15:
        payload = {"player_a": a, "player_b": b >> 2, "ore_type": b & 3}
      if t == 16:
        payload = {"gifter_index": a & 15, "receipient_index": b & 15,
                   "source_type": a >> 4, "received_amount": b >> 4}
        if agent_roles:
          payload.update(gifter_role=agent_roles[(a & 15) - 1],
                         receipient_role=agent_roles[(b & 15) - 1])
      if t == 20:
        payload = {"player_index": a, "mushroom_type": MUSHROOM_TYPES[b - 1]}
      if t in (17, 18, 19):
        payload = {"player_index": a, keys[1]: COOKING_ITEMS[b],
                   **({"receiver": "Receiver"} if t == 17 else {"pot": "CookingPot"})}
      if t == 11 and interaction is not None:
        rewards, inventories = interaction
        payload.update(row_reward=float(rewards[a - 1, 0]), col_reward=float(rewards[a - 1, 1]),
                       row_inventory=inventories[a - 1, 0].copy(),
                       col_inventory=inventories[b - 1, 0].copy())
      out.append((name, payload))
    return out

  # -- buffers -------------------------------------------------------------
  def empty(self, kind: int):
    shape, dtype = self.shapes[kind]
    return self._torch.empty(shape, dtype=dtype, device=self.device)

  def _wrap(self, kind: int, ptr: int, leading: int = 0):
    """A tensor of `kind`'s shape (with `leading` slots in front, if any) over
    engine-library memory (mp_alloc_output / mp_place_output); the memory is released
    (mp_free_output) with the tensor."""
    t = self._torch
    shape, dtype = self.shapes[kind]
    if leading:
      shape = (leading,) + tuple(shape)
    nbytes = int(np.prod(shape)) * t.empty((), dtype=dtype).element_size()
    L, dev_index = self._L, self.device.index or 0

    class _Owner:   # torch keeps this object alive for as long as the tensor's storage
      __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1",
                                  "data": (ptr, False), "version": 2}

      def __del__(self):
        L.mp_free_output(dev_index, ctypes.c_void_p(ptr))

    flat = t.as_tensor(_Owner(), device=self.device)
    return flat.view(dtype).view(shape)

  def empty_mapped(self, kind: int, chunk_bytes: int):
    """A tensor for `kind` whose memory is one virtual range mapped onto separate
    physical chunks of `chunk_bytes` (mp_alloc_output; outside torch's allocator).
    None if the driver refuses."""
    shape, dtype = self.shapes[kind]
    nbytes = int(np.prod(shape)) * self._torch.empty((), dtype=dtype).element_size()
    ptr = ctypes.c_void_p()
    if self._L.mp_alloc_output(self.device.index or 0, nbytes, chunk_bytes, ctypes.byref(ptr)) != 0:
      return None
    try:
      return self._wrap(kind, ptr.value)
    except (RuntimeError, TypeError):
      self._L.mp_free_output(self.device.index or 0, ptr)
      return None

  # A pixel view of at least this many bytes is PLACED (see `place`), not just allocated
  PLACE_MIN_BYTES = 64 << 20

  def place(self, kind: int, candidates: Optional[int] = None, max_bytes: int = 0):
    """Allocates AND BINDS the tensor of a pixel view where this engine's launch
    writes it fastest (mp_place_output: up to `candidates` buffers mapped from 2 MB
    physical chunks, never more than `max_bytes` alive — 0: a quarter of the free
    memory —, each timed with dry launches under the plan that suits it; the fastest
    kept, the others released before the call returns).  The same launch takes
    99 - 122 us (clean_up WORLD.RGB) depending on WHERE its output lies, a property
    of the buffer's physical pages (profiles/r04_write_fronts.md).  What was measured
    stays in `self.placement[kind]`.  A caller that brings its own tensor to `bind`
    gets the speed of that tensor (and the plan tuned to it: mp_tune)."""
    k = self.placements if candidates is None else candidates
    max_bytes = max_bytes or self.place_max_bytes
    ptr, rep = ctypes.c_void_p(), MpPlacement()
    _check(self._L, self._L.mp_place_output(self._h, kind, k, max_bytes, ctypes.byref(ptr),
                                            ctypes.byref(rep)), "mp_place_output")
    try:
      tensor = self._wrap(kind, ptr.value)
    except Exception:
      # the placed buffer must not stay bound (and mapped) behind a failed wrap
      self._L.mp_bind_output(self._h, kind, None)
      self._L.mp_free_output(self.device.index or 0, ptr)
      raise
    self._bound[kind] = tensor
    self.placement[kind] = {"candidates": rep.candidates, "requested": rep.requested,
                            "picked": rep.picked,
                            "dry_launch_us": [round(rep.us[i], 1) for i in range(rep.candidates)],
                            # (every fourth candidate is one plain allocation: mp_place_output)
                            "kind": "plain allocation" if rep.picked % 4 == 3 else "mapped 2 MB",
                            "probe": "stepped behind a copy" if rep.stepped else "dry",
                            "out_of_memory": rep.out_of_memory,
                            "early_exit": {0: None, 1: "round within 3 %",
                                           2: "outlier found"}.get(rep.early_exit),
                            "setup_s": round(rep.setup_ms / 1e3, 3)}
    return tensor

  def tune(self) -> float:
    """The launch plan that suits the pixel views bound right now (mp_tune); returns
    its dry-launch time in us (0.0 when nothing is fused)."""
    us = ctypes.c_double()
    _check(self._L, self._L.mp_tune(self._h, ctypes.byref(us)), "mp_tune")
    return us.value

  def bind(self, kind: int, tensor=None):
    """Binds (and returns) a tensor refreshed by every reset()/step().  Without a
    tensor the engine allocates one — a large pixel view through `place` (unless
    `self.placements` <= 1).  A large pixel view the caller brings gets the launch
    plan tuned to it (mp_tune: a few dry launches)."""
    shape, dtype = self.shapes[kind]
    big = (kind in (OBS_RGB, OBS_WORLD_RGB) and
           int(np.prod(shape)) >= self.PLACE_MIN_BYTES)
    if tensor is None:
      if big and self.placements > 1:
        try:
          return self.place(kind)
        except EngineError as e:
          # no virtual-memory mappings to be had (driver, fragmentation, the bound on
          # retired address space): an ordinary allocation with its plan tuned — slower
          # on most boxes, never a failure to bind
          self.placement[kind] = {"candidates": 0, "kind": "torch allocation (placing failed)",
                                  "error": str(e)}
      tensor = self.empty(kind)
    assert tuple(tensor.shape) == shape and tensor.dtype == dtype
    assert tensor.is_contiguous() and tensor.device == self.device
    _check(self._L, self._L.mp_bind_output(self._h, kind, tensor.data_ptr()),
           "mp_bind_output")
    self._bound[kind] = tensor
    if big and self.placements > 0:
      _check(self._L, self._L.mp_tune(self._h, None), "mp_tune")
    return tensor

  def bind_ring(self, kind: int, tensor=None, slots: Optional[int] = None, tune: bool = True):
    """A rollout ring for `kind` (mp_bind_output_ring): submission t since the ring was
    bound — every reset() and step() — writes slot t % T of `tensor` [T, *shape(kind)].
    All ring-bound kinds share T and the position, so slot s of every kind is the same
    step.  The learner keeps what it was handed: a slot is not written again for T
    submissions, nothing is cloned, and moving on a slot costs a pointer store.  Without
    a tensor one is allocated (`slots` = T; a large pixel view from scattered 2 MB chunks,
    `empty_ring`).  `tune`: time the launch plans on every slot of a large pixel view now
    (once; the timed launches draw into the slots).  (Round 5 also searched a fast set of
    chunks for EVERY slot — 32 slots, 215 candidate sets, 30 s of set-up for 2 %: removed.)"""
    shape, dtype = self.shapes[kind]
    t = self._torch
    if tensor is None:
      if not slots or slots < 1:
        raise ValueError("bind_ring needs a tensor or a positive number of slots")
      tensor = self.empty_ring(kind, int(slots))
    if tuple(tensor.shape[1:]) != tuple(shape) or tensor.dtype != dtype:
      raise ValueError(f"a ring for kind {kind} is [T, {', '.join(map(str, shape))}] {dtype}, "
                       f"got {tuple(tensor.shape)} {tensor.dtype}")
    if slots is not None and int(slots) != tensor.shape[0]:
      raise ValueError(f"{tensor.shape[0]} slots in the tensor, {slots} asked for")
    if not tensor[0].is_contiguous() or tensor.device != self.device:
      raise ValueError("the slots of a ring must be contiguous and on the engine's device")
    # (one slot: any stride that holds it)
    stride = tensor.stride(0) * tensor.element_size() if tensor.shape[0] > 1 else (
        -(-int(np.prod(shape)) * tensor.element_size() // 256) * 256)
    if stride % 256:
      raise ValueError(f"slot stride {stride} B is not a multiple of 256: pad the kind's last axes "
                       "or use Engine.empty_ring")
    _check(self._L, self._L.mp_bind_output_ring(self._h, kind, tensor.data_ptr(), stride,
                                                tensor.shape[0]), "mp_bind_output_ring")
    self._bound[kind] = tensor
    big = kind in (OBS_RGB, OBS_WORLD_RGB) and int(np.prod(shape)) >= self.PLACE_MIN_BYTES
    if tune and big and self.placements > 0:
      _check(self._L, self._L.mp_tune(self._h, None), "mp_tune")
    return tensor

  def empty_ring(self, kind: int, slots: int):
    """[slots, *shape(kind)] whose slots start 256 bytes apart-aligned (the stride
    mp_bind_output_ring wants): for the kinds whose bytes per slot are not a multiple of
    256 the tensor is a view into a padded allocation."""
    shape, dtype = self.shapes[kind]
    t = self._torch
    item = t.empty((), dtype=dtype).element_size()
    n = int(np.prod(shape))
    if (kind in (OBS_RGB, OBS_WORLD_RGB) and n * item >= self.PLACE_MIN_BYTES and
        self.placements > 0 and (n * item) % 256 == 0):
      # a large pixel view: scattered 2 MB chunks, like a placed single buffer — the frame
      # launch writes those evenly; an ordinary allocation is physically contiguous in large
      # pieces and 10 - 15 % slower on about half the boxes (profiles/r05_alloc_method.md)
      ptr = ctypes.c_void_p()
      if self._L.mp_alloc_output(self.device.index or 0, n * item * int(slots), 2 << 20,
                                 ctypes.byref(ptr)) == 0:
        try:
          return self._wrap(kind, ptr.value, leading=int(slots))
        except (RuntimeError, TypeError):
          self._L.mp_free_output(self.device.index or 0, ptr)
    padded = -(-n * item // 256) * 256 // item
    flat = t.empty((int(slots), padded), dtype=dtype, device=self.device)
    return flat[:, :n].view((int(slots),) + tuple(shape)) if padded != n else flat.view(
        (int(slots),) + tuple(shape))

  @property
  def ring(self) -> Dict[str, int]:
    """{"slots": T (0: no ring), "next": the slot the next reset()/step() writes,
    "last": the slot written last}."""
    info = MpInfo()
    _check(self._L, self._L.mp_info(self._h, ctypes.byref(info)), "mp_info")
    T = info.ring_slots
    return {"slots": T, "next": info.ring_next, "last": (info.ring_next + T - 1) % T if T else 0}

  @property
  def retired_va(self) -> Dict[str, int]:
    """Address space this process has retired with released mapped views, and the bound."""
    info = MpInfo()
    _check(self._L, self._L.mp_info(self._h, ctypes.byref(info)), "mp_info")
    return {"bytes": info.retired_va_bytes, "limit": info.retired_va_limit}

  def unbind(self, kind: int):
    _check(self._L, self._L.mp_bind_output(self._h, kind, None),
           "mp_bind_output")
    self._bound.pop(kind, None)

  def use_current_stream(self):
    s = self._torch.cuda.current_stream(self.device).cuda_stream
    _check(self._L, self._L.mp_set_stream(self._h, s), "mp_set_stream")

  # -- episode control -----------------------------------------------------
  def reset(self, seeds: Optional[Sequence[int]] = None, mask=None):
    sp = mp = None
    if seeds is not None:
      seeds = np.ascontiguousarray(seeds, np.uint64)
      assert seeds.shape == (self.N,)
      sp = seeds.ctypes.data
    if mask is not None:
      mask = np.ascontiguousarray(mask, np.uint8)
      assert mask.shape == (self.N,)
      mp = mask.ctypes.data
    _check(self._L, self._L.mp_reset(self._h, sp, mp), "mp_reset")

  def step(self, actions):
    """actions: int32 cuda tensor [N, P] of discrete action ids."""
    t = self._torch
    if isinstance(actions, t.Tensor) and actions.is_cuda:
      assert actions.dtype == t.int32 and actions.is_contiguous()
      assert tuple(actions.shape) == (self.N, self.P)
      _check(self._L, self._L.mp_step(self._h, actions.data_ptr()), "mp_step")
    else:
      a = np.ascontiguousarray(actions, np.int32)
      if a.shape != (self.N, self.P):
        raise ValueError(f"actions must have shape {(self.N, self.P)}")
      _check(self._L, self._L.mp_step_host(self._h, a.ctypes.data),
             "mp_step_host")

  def step_fields(self, fields):
    """The raw action surface of dmlab2d: `fields` int32 [N, P, A], one value per
    field of the avatar's actionOrder (A = info.num_action_fields) — a cuda tensor
    (mp_step_fields) or a host array (mp_step_fields_host, ranges validated)."""
    t = self._torch
    shape = (self.N, self.P, self.info.num_action_fields)
    if isinstance(fields, t.Tensor) and fields.is_cuda:
      assert fields.dtype == t.int32 and fields.is_contiguous()
      assert tuple(fields.shape) == shape
      _check(self._L, self._L.mp_step_fields(self._h, fields.data_ptr()), "mp_step_fields")
    else:
      a = np.ascontiguousarray(fields, np.int32)
      if a.shape != shape:
        raise ValueError(f"fields must have shape {shape}")
      _check(self._L, self._L.mp_step_fields_host(self._h, a.ctypes.data),
             "mp_step_fields_host")

  def observe(self, kind: int, out=None):
    if out is None:
      out = self.empty(kind)
    _check(self._L, self._L.mp_observe(self._h, kind, out.data_ptr()),
           "mp_observe")
    return out

  def observe_host(self, kind: int) -> np.ndarray:
    """Observation `kind` of all worlds as a host array (synchronises)."""
    return self.observe(kind).cpu().numpy()

  # -- introspection -------------------------------------------------------
  def dump(self):
    i = self.info
    grid = np.zeros((self.N, i.num_layers, i.map_h, i.map_w), np.uint8)
    avat = np.zeros((self.N, self.P, 8), np.int32)
    glob = np.zeros((self.N, 8), np.int32)
    _check(self._L, self._L.mp_dump(self._h, grid.ctypes.data,
                                    avat.ctypes.data, glob.ctypes.data),
           "mp_dump")
    return grid, avat, glob

  def snapshot(self) -> np.ndarray:
    n = int(self._L.mp_snapshot_bytes(self._h))
    buf = np.zeros(n, np.uint8)
    _check(self._L, self._L.mp_snapshot(self._h, buf.ctypes.data, n),
           "mp_snapshot")
    return buf

  def restore(self, buf: np.ndarray):
    buf = np.ascontiguousarray(buf, np.uint8)
    _check(self._L, self._L.mp_restore(self._h, buf.ctypes.data, buf.size),
           "mp_restore")

  def counters(self) -> Dict[str, int]:
    out = np.zeros(len(COUNTER_NAMES), np.uint64)
    _check(self._L, self._L.mp_counters(self._h, out.ctypes.data),
           "mp_counters")
    # (reward_sum_x1024 is a signed sum: coins pays negative rewards)
    return {k: int(v.astype(np.int64)) if k == "reward_sum_x1024" else int(v)
            for k, v in zip(COUNTER_NAMES, out)}

  def sync(self):
    _check(self._L, self._L.mp_sync(self._h), "mp_sync")

  def box_fill(self, kind: int, reps: int = 20):
    """mp_box_fill: what the box's memory system gives the buffer bound for pixel view
    `kind` — µs per launch of the runtime's memset, of a bare store loop in the frame
    launch's write order and of the same bytes as one chip-wide 4 KiB front.  OVERWRITES
    the view with junk (the next step redraws it)."""
    rep = MpBoxFill()
    _check(self._L, self._L.mp_box_fill(self._h, int(kind), int(reps), ctypes.byref(rep)), "mp_box_fill")
    return {"bytes": int(rep.bytes), "memset_us": round(rep.memset_us, 2),
            "product_order_us": round(rep.product_order_us, 2),
            "front_4k_us": round(rep.front_4k_us, 2), "workgroups": rep.groups,
            "storing_waves": rep.waves, "span_bytes": int(rep.span_bytes)}

  def fault_words(self) -> np.ndarray:
    """Diagnostics of the frame kernel's pipeline (include/mp_engine.h); host
    memory, never blocks."""
    out = np.zeros(64, np.uint32)
    _check(self._L, self._L.mp_fault_words(self._h, out.ctypes.data), "mp_fault_words")
    return out


def pack_agent_roles(pack_bytes: bytes):
  """The avatars' `agentRole` strings (gift_refinements: what the `gift` event reports as
  gifter_role / receipient_role, components.lua:174-181), () for a pack without them."""
  from meltingpot_amd import pack as pack_lib
  t = pack_lib.loads(pack_bytes)
  if "agent_roles" not in t:
    return ()
  return tuple(n.decode() for n in bytes(t["agent_roles"]).split(b"\0")[:-1])


def pack_role_names(pack_bytes: bytes):
  """Names of the roles a pack carries per-player constants for (sorted; the
  index is what MpConfig.roles takes), or None for a single-role substrate."""
  from meltingpot_amd import pack as pack_lib
  t = pack_lib.loads(pack_bytes)
  if "role_names" not in t:
    return None
  return tuple(n.decode() for n in bytes(t["role_names"]).split(b"\0")[:-1])


def load_pack(name: str) -> bytes:
  """Committed lowered-substrate pack (generated by tools/make_packs.py)."""
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets",
                      f"{name}.mpk")
  with open(path, "rb") as f:
    return f.read()
