"""The reference's own per-substrate conformance test — meltingpot/substrate_test.py:24-47,
`PerSubstrateTestCase`, parametrised over `substrate.SUBSTRATES`, with
meltingpot/testing/substrates.py:22-68 `assert_step_matches_specs` — restated for EVERY
substrate this package registers: `get_factory(name)`, the default roles, the env's
action / reward / discount / observation specs against the factory's, one step with the
largest action of every player validated leaf by leaf.

CPU: the product's `Substrate` on the oracle (`oracle_engine.OracleBatchEngine`, injected by the
test); GPU: the same on the HIP engine, plus a batched build of every substrate.  With the
reference tree at hand, every registered name's roles, specs and action set are held against
the reference's config module of that name."""
import os

import numpy as np
import pytest

import util  # noqa: F401
from meltingpot_amd import substrate

HAVE_REFERENCE = os.path.isdir("/root/reference/meltingpot")
NAMES = sorted(substrate.SUBSTRATES)


def _assert_step_matches_specs(env):
  """meltingpot/testing/substrates.py:22-68."""
  env.reset()
  action = [int(spec.maximum) for spec in env.action_spec()]
  timestep = env.step(action)
  env.discount_spec().validate(np.float64(timestep.discount))
  reward_spec = env.reward_spec()
  assert len(reward_spec) == len(timestep.reward)
  for n, spec in enumerate(reward_spec):
    spec.validate(timestep.reward[n])
  observation_specs = env.observation_spec()
  assert len(observation_specs) == len(timestep.observation)
  for n, (observation, spec) in enumerate(zip(timestep.observation, observation_specs)):
    assert set(spec) == set(observation), n
    for key in spec:
      spec[key].validate(observation[key])


def _per_substrate(name):
  """meltingpot/substrate_test.py:27-47."""
  factory = substrate.get_factory(name)
  roles = factory.default_player_roles()
  action_spec = [factory.action_spec()] * len(roles)
  reward_spec = [factory.timestep_spec().reward] * len(roles)
  discount_spec = factory.timestep_spec().discount
  observation_spec = dict(factory.timestep_spec().observation)
  observation_spec["COLLECTIVE_REWARD"] = substrate.Array((), np.float64, "COLLECTIVE_REWARD")
  observation_spec = [observation_spec] * len(roles)
  with factory.build(roles) as env:
    _assert_step_matches_specs(env)
    assert list(env.action_spec()) == action_spec
    assert list(env.reward_spec()) == reward_spec
    assert env.discount_spec() == discount_spec
    assert list(env.observation_spec()) == observation_spec


def test_the_registry_is_the_committed_packs():
  assets = os.path.join(os.path.dirname(os.path.abspath(substrate.__file__)), "assets")
  assert {f[:-4] for f in os.listdir(assets) if f.endswith(".mpk")} == set(NAMES)
  assert len(NAMES) == 33


@pytest.mark.parametrize("name", NAMES)
def test_substrate_on_the_oracle(name, monkeypatch):
  from oracle_engine import OracleBatchEngine
  monkeypatch.setattr(substrate.engine_lib, "Engine", OracleBatchEngine)
  _per_substrate(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_substrate_on_the_hip_engine(name):
  _per_substrate(name)
  # ... and batched: 5 worlds in one build, leaves with a leading [5], three steps
  import torch
  factory = substrate.get_factory(name)
  roles = factory.default_player_roles()
  with substrate.build(name, roles=roles, num_worlds=5) as env:
    ts = env.reset()
    nact = factory.action_spec().num_values
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    for _ in range(3):
      acts = torch.randint(0, nact, (5, len(roles)), generator=gen, device="cuda", dtype=torch.int32)
      ts = env.step(acts)
    spec = factory.timestep_spec().observation
    for key, sp in spec.items():
      leaf = ts.observation[key]
      if key.startswith("WORLD."):
        assert tuple(leaf.shape) == (5,) + tuple(sp.shape), key
      else:
        assert tuple(leaf.shape) == (5, len(roles)) + tuple(sp.shape), key
    assert tuple(ts.reward.shape) == (5, len(roles))


@pytest.mark.skipif(not HAVE_REFERENCE, reason="no reference tree on this box")
@pytest.mark.parametrize("name", NAMES)
def test_registered_config_is_the_reference_config(name):
  """configs/substrates/<name>.py:get_config(): roles, per-player timestep spec, action set."""
  from meltingpot_amd import refshim
  ref = refshim.load_config_module(name).get_config()
  f = substrate.get_factory(name)
  cfg = substrate.get_config(name)
  assert f.valid_roles() == frozenset(ref.valid_roles)
  assert tuple(f.default_player_roles()) == tuple(ref.default_player_roles)
  ref_obs = ref.timestep_spec if isinstance(ref.timestep_spec, dict) else ref.timestep_spec.observation
  ours = f.timestep_spec().observation
  assert set(ours) == set(ref_obs)
  for k, sp in ref_obs.items():
    assert tuple(ours[k].shape) == tuple(sp.shape) and np.dtype(ours[k].dtype) == np.dtype(sp.dtype), k
  assert f.action_spec().num_values == len(ref.action_set)
  assert [dict(a) for a in cfg.action_set] == [dict(a) for a in ref.action_set]
  assert list(cfg.individual_observation_names) == list(ref.individual_observation_names)
  assert list(cfg.global_observation_names) == list(ref.global_observation_names)


@pytest.mark.skipif(not HAVE_REFERENCE, reason="no reference tree on this box")
@pytest.mark.parametrize("name", NAMES)
def test_reference_config_object_lowered_at_run_time(name, monkeypatch):
  """meltingpot/substrate.py:98-113 with the REFERENCE's config object of every registered
  substrate (configs/substrates/__init__.py:58-67 attaches `lab2d_settings_builder`): the
  factory lowers the settings the config builds at run time — no committed pack involved —
  and the substrate passes the same conformance check and shows the world the committed
  pack shows for the same seed and actions (coins draws its map with Python's `random`
  inside build(): not compared)."""
  import random
  from oracle_engine import OracleBatchEngine
  from meltingpot_amd import refshim
  monkeypatch.setattr(substrate.engine_lib, "Engine", OracleBatchEngine)
  mod = refshim.load_config_module(name)
  config = mod.get_config()
  with config.unlocked():
    config.lab2d_settings_builder = lambda *, roles, config: mod.build(roles, config)
    config.action_spec = substrate.DiscreteArray(len(config.action_set))
    config.timestep_spec = substrate.timestep_spec_of({
        k: substrate.Array(v.shape, v.dtype, k) for k, v in dict(config.timestep_spec).items()})
  roles = tuple(config.default_player_roles)
  random.seed(0)
  factory = substrate.get_factory_from_config(config)
  assert factory.valid_roles() == frozenset(config.valid_roles)
  with factory.build(roles, env_seed=11) as env:
    _assert_step_matches_specs(env)
  if name == "coins":
    return
  # (fresh substrates: episode e of a world is keyed by (seed, e))
  random.seed(0)
  with factory.build(roles, env_seed=11) as env, substrate.build(name, roles=roles, env_seed=11) as packed:
    a, b = env.reset(), packed.reset()
    for step in range(4):
      assert np.array_equal(a.observation[0]["WORLD.RGB"], b.observation[0]["WORLD.RGB"]), step
      assert a.reward == b.reward
      action = [(step + i) % int(spec.num_values) for i, spec in enumerate(env.action_spec())]
      a, b = env.step(action), packed.step(action)
