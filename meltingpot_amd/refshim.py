"""Load a reference substrate config module without the reference's heavy deps.

The reference config modules (`meltingpot/configs/substrates/<name>.py`) are
plain Python: the ASCII map, char->prefab map, prefab dicts, avatar builders and
`build(roles, config)` need nothing but `shapes.py` / `colors.py`.  Only
`get_config()` touches `ml_collections` / `dm_env`.  This shim installs stub
modules for those so `build()` can be executed in a container that has neither
("the intended stage-0 ingestion route").

If a real `meltingpot` package is importable it is used instead.

This module is only needed to (re)generate `meltingpot_amd/assets/*.mpk`, by
the CPU test that checks the committed packs are up to date, and by the
conformance test that layers the reference's own wrapper stack on
`lab2d_env.Environment` (`load_reference_wrappers`).  Nothing at run time on the
GPU box imports it.
"""

from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types
from typing import Any, Mapping, Sequence

DEFAULT_REFERENCE_ROOT = os.environ.get("MELTINGPOT_REFERENCE_ROOT",
                                        "/root/reference")


class _ConfigDict(dict):
  """Minimal stand-in for ml_collections.ConfigDict (attribute access)."""

  def __getattr__(self, k):
    try:
      v = self[k]
    except KeyError as e:
      raise AttributeError(k) from e
    if isinstance(v, dict) and not isinstance(v, _ConfigDict):
      v = _ConfigDict(v)   # ml_collections wraps nested dicts the same way
      self[k] = v
    return v

  def __setattr__(self, k, v):
    self[k] = v

  def lock(self):
    return self

  def unlock(self):
    return self

  def to_dict(self):
    return {k: (v.to_dict() if isinstance(v, _ConfigDict) else v) for k, v in self.items()}

  def unlocked(self):
    import contextlib
    return contextlib.nullcontext(self)


class _Spec:
  """Duck-typed dm_env.specs.Array good enough for get_config()."""

  def __init__(self, shape=(), dtype="float64", name=None, num_values=None):
    self.shape = tuple(shape)
    self.dtype = dtype
    self.name = name
    self.num_values = num_values

  def replace(self, **kw):
    d = dict(shape=self.shape, dtype=self.dtype, name=self.name,
             num_values=self.num_values)
    d.update(kw)
    return _Spec(**d)

  def __repr__(self):
    return f"Spec(shape={self.shape}, dtype={self.dtype}, name={self.name})"


def _load(path: str, name: str):
  spec = importlib.util.spec_from_file_location(name, path)
  module = importlib.util.module_from_spec(spec)
  sys.modules[name] = module
  spec.loader.exec_module(module)
  return module


def _install_stubs(root: str) -> None:
  if "ml_collections" not in sys.modules:
    try:
      importlib.import_module("ml_collections")
    except ImportError:
      ml = types.ModuleType("ml_collections")
      cd = types.ModuleType("ml_collections.config_dict")
      cd.ConfigDict = _ConfigDict
      ml.config_dict = cd
      ml.ConfigDict = _ConfigDict
      sys.modules["ml_collections"] = ml
      sys.modules["ml_collections.config_dict"] = cd

  for pkg in ("meltingpot", "meltingpot.utils", "meltingpot.utils.substrates",
              "meltingpot.configs", "meltingpot.configs.substrates"):
    if pkg not in sys.modules:
      m = types.ModuleType(pkg)
      m.__path__ = []
      sys.modules[pkg] = m
  # config modules may import a sibling base module (territory__rooms imports
  # territory): let the normal import machinery find plain .py siblings
  sys.modules["meltingpot.configs.substrates"].__path__ = [
      os.path.join(root, "meltingpot", "configs", "substrates")]
  sys.modules["meltingpot.configs"].substrates = sys.modules[
      "meltingpot.configs.substrates"]

  base = os.path.join(root, "meltingpot", "utils", "substrates")
  mus = sys.modules["meltingpot.utils.substrates"]
  for leaf in ("colors", "shapes", "map_helpers"):
    full = f"meltingpot.utils.substrates.{leaf}"
    if full not in sys.modules:
      _load(os.path.join(base, f"{leaf}.py"), full)
    setattr(mus, leaf, sys.modules[full])

  full = "meltingpot.utils.substrates.specs"
  if full not in sys.modules:
    specs = types.ModuleType(full)
    specs.OBSERVATION = {
        "RGB": _Spec((88, 88, 3), "uint8", "RGB"),
        "READY_TO_SHOOT": _Spec((), "float64", "READY_TO_SHOOT"),
        "POSITION": _Spec((2,), "int32", "POSITION"),
        "ORIENTATION": _Spec((), "int32", "ORIENTATION"),
    }
    specs.float32 = lambda *s, name=None: _Spec(s, "float32", name)
    specs.float64 = lambda *s, name=None: _Spec(s, "float64", name)
    specs.int32 = lambda *s, name=None: _Spec(s, "int32", name)
    specs.int64 = lambda *s, name=None: _Spec(s, "int64", name)
    specs.action = lambda n: _Spec((), "int64", "action", num_values=n)
    specs.rgb = lambda h, w, name="RGB": _Spec((h, w, 3), "uint8", name)

    def world_rgb(ascii_map, sprite_size, name="WORLD.RGB"):
      lines = ascii_map.strip().split("\n")
      return _Spec((len(lines) * sprite_size, len(lines[0]) * sprite_size, 3),
                   "uint8", name)

    specs.world_rgb = world_rgb
    specs.inventory = lambda n, name="INVENTORY": _Spec((n,), "float64", name)
    specs.interaction_inventories = (
        lambda n, name="INTERACTION_INVENTORIES": _Spec((2, n), "float64",
                                                        name))
    specs.timestep = lambda obs: dict(obs)
    sys.modules[full] = specs
  mus.specs = sys.modules[full]

  # game_object_utils only needs colors, shapes and numpy: load the real one
  # (coins builds its avatars with build_avatar_objects)
  full = "meltingpot.utils.substrates.game_object_utils"
  if full not in sys.modules:
    _load(os.path.join(base, "game_object_utils.py"), full)
  mus.game_object_utils = sys.modules[full]


def load_config_module(name: str, root: str = DEFAULT_REFERENCE_ROOT):
  """Returns the reference config module `configs/substrates/<name>.py`."""
  full = f"meltingpot.configs.substrates.{name}"
  if full in sys.modules:
    return sys.modules[full]
  path = os.path.join(root, "meltingpot", "configs", "substrates",
                      f"{name}.py")
  if not os.path.exists(path):
    raise FileNotFoundError(
        f"reference config {path} not found (set MELTINGPOT_REFERENCE_ROOT)")
  _install_stubs(root)
  # territory__rooms etc. import their base module (territory).
  return _load(path, full)


def build_settings(name: str, roles: Sequence[str],
                   root: str = DEFAULT_REFERENCE_ROOT) -> Mapping[str, Any]:
  """Runs the reference `build(roles, config)` and returns the lab2d settings
  dict (reference: configs/substrates/clean_up.py:841-865)."""
  module = load_config_module(name, root)
  config = module.get_config()
  return module.build(tuple(roles), config), module, config


# --------------------------------------------------------------------------
# The reference's wrapper stack, loaded UNMODIFIED from the reference tree.


def _stub(name: str, **attrs):
  """Installs module `name` with `attrs` unless the real one imports."""
  m = sys.modules.get(name)
  if m is None:
    try:
      return importlib.import_module(name)
    except ImportError:
      pass
    m = types.ModuleType(name)
    m.__path__ = []
    m._mp_stub = True
    sys.modules[name] = m
  if getattr(m, "_mp_stub", False):
    for k, v in attrs.items():
      setattr(m, k, v)
  if "." in name:
    parent, leaf = name.rsplit(".", 1)
    setattr(_stub(parent), leaf, m)
  return m


def _install_wrapper_stubs() -> None:
  """Stand-ins for the third-party packages the wrapper modules import
  (dm_env, dmlab2d, reactivex, chex, immutabledict, absl, tree): exactly the
  names those files use, backed by this package's own spec / timestep / subject
  classes, so that `isinstance(x, dm_env.specs.Array)` and friends mean the
  same objects on both sides."""
  import dataclasses
  import unittest
  from meltingpot_amd import substrate as ours

  class _Environment:   # dm_env.Environment / dmlab2d.Environment: an interface
    def close(self):
      pass

    def __enter__(self):
      return self

    def __exit__(self, *exc):
      self.close()

  specs = _stub("dm_env.specs", Array=ours.Array, BoundedArray=ours.BoundedArray,
                DiscreteArray=ours.DiscreteArray)
  _stub("dm_env", TimeStep=ours.TimeStep, StepType=ours.StepType, Environment=_Environment,
        specs=specs)
  dm_env = sys.modules["dm_env"]
  _stub("dmlab2d", Environment=getattr(dm_env, "Environment", _Environment))
  _stub("dmlab2d.runfiles_helper", find=lambda: "")   # builder.py:35 (module level)
  _stub("dmlab2d.settings_helper")

  class _Observable:
    def __class_getitem__(cls, item):
      return cls

  def _empty():
    done = ours.Subject()
    done.on_completed()
    return done

  def _map(fn):
    def operator(source):
      out = ours.Subject()
      source.subscribe(on_next=lambda v: out.on_next(fn(v)), on_completed=out.on_completed)
      return out
    return operator

  _stub("reactivex", Observable=_Observable, empty=_empty)
  _stub("reactivex.subject", Subject=ours.Subject)
  _stub("reactivex.operators", map=_map)

  def _chex_dataclass(cls=None, **kw):
    kw.pop("mappable_dataclass", None)
    wrap = lambda c: dataclasses.dataclass(c, **kw)
    return wrap if cls is None else wrap(cls)

  _stub("chex", dataclass=_chex_dataclass)
  _stub("immutabledict", immutabledict=lambda *a, **kw: types.MappingProxyType(dict(*a, **kw)))
  import logging as _logging
  _stub("absl")
  _stub("absl.logging", info=_logging.info, warning=_logging.warning, error=_logging.error)
  _stub("absl.testing")
  _stub("absl.testing.parameterized", TestCase=unittest.TestCase)
  _stub("tree")
  _install_stubs(DEFAULT_REFERENCE_ROOT)   # ml_collections + the meltingpot package shells


def load_reference_wrappers(root: str = DEFAULT_REFERENCE_ROOT):
  """Imports, from the reference tree and without touching them, the modules of
  `build_substrate`'s wrapper stack (utils/substrates/substrate.py:107-139) and
  the conformance helper (testing/substrates.py).  Returns a namespace with
  `base`, `observables`, `observables_wrapper`, `multiplayer_wrapper`,
  `discrete_action_wrapper`, `collective_reward_wrapper`, `substrate` and
  `testing_substrates`."""
  _install_wrapper_stubs()
  base_dir = os.path.join(root, "meltingpot", "utils", "substrates")
  for pkg in ("meltingpot.utils.substrates.wrappers", "meltingpot.testing"):
    if pkg not in sys.modules:
      m = types.ModuleType(pkg)
      m.__path__ = []
      sys.modules[pkg] = m
  wr = sys.modules["meltingpot.utils.substrates.wrappers"]
  sys.modules["meltingpot.utils.substrates"].wrappers = wr
  out = types.SimpleNamespace()
  for leaf in ("base", "observables", "observables_wrapper", "reset_wrapper",
               "multiplayer_wrapper", "discrete_action_wrapper", "collective_reward_wrapper"):
    full = f"meltingpot.utils.substrates.wrappers.{leaf}"
    if full not in sys.modules:
      _load(os.path.join(base_dir, "wrappers", f"{leaf}.py"), full)
    setattr(wr, leaf, sys.modules[full])
    setattr(out, leaf, sys.modules[full])
  for leaf in ("builder", "substrate"):
    full = f"meltingpot.utils.substrates.{leaf}"
    if full not in sys.modules:
      _load(os.path.join(base_dir, f"{leaf}.py"), full)
    setattr(sys.modules["meltingpot.utils.substrates"], leaf, sys.modules[full])
  out.substrate = sys.modules["meltingpot.utils.substrates.substrate"]
  full = "meltingpot.testing.substrates"
  if full not in sys.modules:
    _load(os.path.join(root, "meltingpot", "testing", "substrates.py"), full)
  out.testing_substrates = sys.modules[full]
  return out


def load_reference_scenarios(root: str = DEFAULT_REFERENCE_ROOT):
  """Imports, from the reference tree and without touching them, what stands ON a
  substrate: `utils/policies/{policy,fixed_action_policy,policy_factory}.py`,
  `utils/scenarios/{population,scenario,scenario_factory}.py` and
  `utils/evaluation/{return_subject,evaluation}.py` (what the
  `Substrate` API is FOR).  Bots (saved models) and videos are out of scope: the two
  modules evaluation.py imports for them are empty stand-ins.  Returns a namespace with
  `policy`, `fixed_action_policy`, `policy_factory`, `population`, `scenario`,
  `scenario_factory`, `return_subject`, `evaluation` (+ everything
  `load_reference_wrappers` returns as `wrappers`)."""
  import concurrent.futures  # noqa: F401  (population.py says `import concurrent`)
  wrappers = load_reference_wrappers(root)
  base = os.path.join(root, "meltingpot", "utils")
  for pkg in ("meltingpot.utils.policies", "meltingpot.utils.scenarios",
              "meltingpot.utils.evaluation"):
    if pkg not in sys.modules:
      m = types.ModuleType(pkg)
      m.__path__ = []
      sys.modules[pkg] = m
    setattr(sys.modules["meltingpot.utils"], pkg.rsplit(".", 1)[1], sys.modules[pkg])
  sys.modules["meltingpot"].utils = sys.modules["meltingpot.utils"]
  out = types.SimpleNamespace(wrappers=wrappers)

  def load(package, leaf):
    full = f"meltingpot.utils.{package}.{leaf}"
    if full not in sys.modules:
      _load(os.path.join(base, package, f"{leaf}.py"), full)
    setattr(sys.modules[f"meltingpot.utils.{package}"], leaf, sys.modules[full])
    setattr(out, leaf, sys.modules[full])

  # (scenario_factory.py names the substrate factory's type: the reference's own module)
  full = "meltingpot.utils.substrates.substrate_factory"
  if full not in sys.modules:
    _load(os.path.join(base, "substrates", "substrate_factory.py"), full)
  sys.modules["meltingpot.utils.substrates"].substrate_factory = sys.modules[full]
  out.substrate_factory = sys.modules[full]
  for leaf in ("policy", "fixed_action_policy", "policy_factory"):
    load("policies", leaf)
  for leaf in ("population", "scenario", "scenario_factory"):
    load("scenarios", leaf)
  # evaluation.py's imports for saved-model bots and videos (tensorflow, cv2): not on this path
  _stub("meltingpot.utils.policies.saved_model_policy")
  _stub("meltingpot.utils.evaluation.video_subject")
  load("evaluation", "return_subject")
  load("evaluation", "evaluation")
  return out
