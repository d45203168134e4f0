"""The reference's OWN wrapper stack — the files under
/root/reference/meltingpot/utils/substrates/wrappers/ and substrate.py, imported
unmodified — layered on `lab2d_env.Environment`, and the reference's own
conformance check (`meltingpot/testing/substrates.py:22-68`) run on the result.

`build_substrate` (utils/substrates/substrate.py:107-139) is

    env = builder.builder(lab2d_settings)              # dmlab2d: replaced
    env = observables_wrapper.ObservablesWrapper(env)
    env = multiplayer_wrapper.Wrapper(env, ...)
    env = discrete_action_wrapper.Wrapper(env, action_table=...)
    env = collective_reward_wrapper.CollectiveRewardWrapper(env)
    return Substrate(env)

and this file builds exactly that with `lab2d_env.Environment` in the first
line.  The third-party packages those files import (dm_env, dmlab2d, reactivex,
chex, immutabledict, absl) are not installed here; `refshim` stands in for them
with this package's own spec / timestep / subject classes.

Here (no GPU, reference tree present) the world behind `lab2d_env.Environment`
is the CPU oracle (tests/oracle_engine.py); on the GPU box (no reference tree)
`tests/test_substrate_api.py::test_flat_lab2d_environment_on_the_hip_engine`
checks that the HIP engine behind the same class yields the same flat timesteps.
"""
import os

import numpy as np
import pytest

import util
from meltingpot_amd import engine, lab2d_env, refshim, substrate
from oracle_engine import OracleEngine

pytestmark = pytest.mark.skipif(not os.path.isdir(refshim.DEFAULT_REFERENCE_ROOT),
                                reason="reference tree not present (GPU box)")


def _reference_stack(name, roles, seed, action_table=None):
  ref = refshim.load_reference_wrappers()
  cfg = substrate.get_config(name)
  raw = lab2d_env.Environment(
      name, roles, engine=OracleEngine(engine.load_pack(name), seed, len(roles)))
  env = ref.observables_wrapper.ObservablesWrapper(raw)
  env = ref.multiplayer_wrapper.Wrapper(
      env, individual_observation_names=cfg.individual_observation_names,
      global_observation_names=cfg.global_observation_names)
  env = ref.discrete_action_wrapper.Wrapper(
      env, action_table=cfg.action_set if action_table is None else action_table)
  env = ref.collective_reward_wrapper.CollectiveRewardWrapper(env)
  return ref, cfg, ref.substrate.Substrate(env)


@pytest.mark.parametrize("name,players", [
    ("clean_up", 7), ("clean_up", 3), ("commons_harvest__open", 7),
    ("commons_harvest__open", 16), ("territory__rooms", 9), ("coins", 2)])
def test_reference_assert_step_matches_specs(name, players):
  """substrate_test.py:24-47 runs `assert_step_matches_specs` on every substrate;
  here it runs — the reference's own method, unmodified — on the reference's own
  wrappers over this repo's environment."""
  ref, cfg, env = _reference_stack(name, ("default",) * players, seed=11)
  case = ref.testing_substrates.SubstrateTestCase()
  with env:
    case.assert_step_matches_specs(env)
    assert len(env.action_spec()) == players
    assert env.action_spec()[0].num_values == len(cfg.action_set)
    assert [s.name for s in env.reward_spec()] == ["REWARD"] * players


@pytest.mark.parametrize("name", [
    "prisoners_dilemma_in_the_matrix__repeated", "running_with_scissors_in_the_matrix__arena",
    "bach_or_stravinsky_in_the_matrix__repeated", "stag_hunt_in_the_matrix__arena",
    "running_with_scissors_in_the_matrix__one_shot"])
def test_reference_assert_step_matches_specs_in_the_matrix(name):
  """The same conformance check on the *_in_the_matrix substrates: INVENTORY and
  INTERACTION_INVENTORIES against the reference configs' own `timestep_spec`
  (specs.inventory / specs.interaction_inventories)."""
  ref_cfg = refshim.load_config_module(name).get_config()
  roles = tuple(ref_cfg.default_player_roles)
  ref, cfg, env = _reference_stack(name, roles, seed=5)
  case = ref.testing_substrates.SubstrateTestCase()
  with env:
    case.assert_step_matches_specs(env)
    # our spec tables are the reference config's
    for key, spec in ref_cfg.timestep_spec.items():
      assert tuple(cfg.timestep_spec[key].shape) == tuple(spec.shape), key
      assert np.dtype(cfg.timestep_spec[key].dtype) == np.dtype(spec.dtype), key
    ts = env.reset()
    for _ in range(5):
      ts = env.step([1] * len(roles))
    assert ts.observation[0]["INVENTORY"].shape == ref_cfg.timestep_spec["INVENTORY"].shape
    assert np.all(ts.observation[0]["INTERACTION_INVENTORIES"] == -1.0)


def test_reference_stack_timesteps_are_the_oracles():
  """30 steps through the reference wrappers: per-player lists, None discount ->
  0., COLLECTIVE_REWARD = sum of rewards (collective_reward_wrapper.py:49) — and
  every leaf equal to what the world underneath produced."""
  from oracle import oracle as oracle_lib
  roles = ("default",) * 7
  ref, cfg, env = _reference_stack("clean_up", roles, seed=util.world_seed(3))
  o = oracle_lib.Oracle(util.fertile_clean_up(engine.load_pack("clean_up")), util.world_seed(3))
  del o   # (the stack below runs the stock pack)
  o = oracle_lib.Oracle(engine.load_pack("clean_up"), util.world_seed(3), 7)
  o.reset()
  ts = env.reset()
  assert ts.step_type == substrate.StepType.FIRST and ts.discount == 0.0
  assert [float(r) for r in ts.reward] == [0.0] * 7
  rng = np.random.default_rng(0)
  for _ in range(30):
    acts = rng.integers(0, 9, 7)
    ts = env.step([int(a) for a in acts])
    o.step(acts.astype(np.int32))
    assert ts.step_type == substrate.StepType.MID and ts.discount == 1.0
    assert [float(r) for r in ts.reward] == list(o.rewards())
    assert len(ts.observation) == 7
    for p, obs in enumerate(ts.observation):
      assert set(obs) == {"RGB", "READY_TO_SHOOT", "NUM_OTHERS_WHO_CLEANED_THIS_STEP",
                          "WORLD.RGB", "COLLECTIVE_REWARD"}
      assert np.array_equal(obs["RGB"], o.render_agent(p))
      assert np.array_equal(obs["WORLD.RGB"], o.render_world())
      assert obs["READY_TO_SHOOT"] == o.ready_to_shoot()[p]
      assert obs["COLLECTIVE_REWARD"] == o.rewards().sum()
  env.close()


def test_reference_stack_validates_actions_and_emits_observables():
  ref, cfg, env = _reference_stack("clean_up", ("default",) * 7, seed=5)
  seen = {"action": [], "timestep": [], "events": [], "done": []}
  obs = env.observables()
  obs.action.subscribe(on_next=seen["action"].append)
  obs.timestep.subscribe(on_next=seen["timestep"].append,
                         on_completed=lambda: seen["done"].append(1))
  obs.events.subscribe(on_next=seen["events"].append)
  obs.dmlab2d.events.subscribe(on_next=lambda e: None)   # the inner observables exist too
  env.reset()
  env.step([8] * 7)   # FIRE_CLEAN
  with pytest.raises(IndexError):      # discrete_action_wrapper.py:99: a plain table lookup
    env.step([9] + [0] * 6)
  env.step([1] * 6)                    # a missing player's actions stay 0, as in dmlab2d
  env.close()
  # discrete_action_wrapper.py:28-49: it is the action TABLE that is validated,
  # against the flat environment's action spec
  _, _, stack = _reference_stack("clean_up", ("default",) * 7, seed=5)
  multiplayer = stack._env._env._env   # Substrate -> CollectiveReward -> DiscreteAction -> Multiplayer
  with pytest.raises(ValueError):
    ref.discrete_action_wrapper.Wrapper(multiplayer, action_table=[{"move": 7}])
  stack.close()
  assert len(seen["action"]) == 3 and len(seen["timestep"]) == 3 and seen["done"] == [1]
  names = [name for name, _ in seen["events"]]
  assert names[:7] == ["AvatarStarted"] * 7


def test_lab2d_environment_has_the_whole_dmlab2d_surface():
  """wrappers/base.py:38-84 forwards exactly these."""
  raw = lab2d_env.Environment("clean_up", ("default",) * 7,
                              engine=OracleEngine(engine.load_pack("clean_up"), 3, 7))
  for method in ("reset", "step", "reward_spec", "discount_spec", "observation_spec",
                 "action_spec", "close", "observation", "events", "list_property",
                 "write_property", "read_property"):
    assert callable(getattr(raw, method)), method
  assert raw.list_property("") == []
  with pytest.raises(KeyError):
    raw.read_property("no.such.property")
  raw.discount_spec().validate(np.float64(1.0))
  raw.reward_spec().validate(np.float64(0.0))
  assert set(raw.action_spec()) == {f"{p}.{k}" for p in range(1, 8)
                                    for k in ("move", "turn", "fireZap", "fireClean")}
  ts = raw.reset()
  assert ts.reward is None and ts.discount is None     # dmlab2d on FIRST
  assert set(ts.observation) == set(raw.observation_spec())
  for k, spec in raw.observation_spec().items():
    spec.validate(ts.observation[k])
  raw.close()


# A table the configs do not have: every row does several things at once (the
# reference accepts any table inside the action spec, discrete_action_wrapper.py:
# 77-109; a human player's step sets move, turn and fire fields independently,
# human_players/level_playing_utils.py:283,333-334).
COMPOSITE_TABLE = (
    {"move": 0, "turn": 0, "fireZap": 0, "fireClean": 0},
    {"move": 1, "turn": 1, "fireZap": 0, "fireClean": 0},     # forward while turning right
    {"move": 3, "turn": -1, "fireZap": 1, "fireClean": 0},    # back, turn left, zap
    {"move": 2, "turn": 0, "fireZap": 0, "fireClean": 1},     # strafe and clean
    {"move": 4, "turn": 1, "fireZap": 1, "fireClean": 1},     # everything
)


def test_reference_discrete_action_wrapper_with_a_custom_table():
  """The reference's discrete_action_wrapper.Wrapper(env, action_table=custom),
  unmodified, on this repo's flat environment: composite rows reach the world as
  raw fields (mp_step_fields) and do what the oracle does with the same fields."""
  from oracle import oracle as oracle_lib
  roles = ("default",) * 7
  ref, cfg, env = _reference_stack("clean_up", roles, seed=util.world_seed(9),
                                   action_table=COMPOSITE_TABLE)
  o = oracle_lib.Oracle(engine.load_pack("clean_up"), util.world_seed(9), 7)
  o.reset()
  env.reset()
  assert env.action_spec()[0].num_values == len(COMPOSITE_TABLE)
  names = ("move", "turn", "fireZap", "fireClean")
  rows = np.array([[r[n] for n in names] for r in COMPOSITE_TABLE], np.int32)
  rng = np.random.default_rng(4)
  moved_and_turned = False
  for _ in range(60):
    acts = rng.integers(0, len(COMPOSITE_TABLE), 7)
    before = o.dump()[1][:, :3].copy()
    ts = env.step([int(a) for a in acts])
    o.step_fields(rows[acts])
    after = o.dump()[1][:, :3]
    moved_and_turned |= bool(np.any((before[:, 2] != after[:, 2]) &
                                    ((before[:, 0] != after[:, 0]) | (before[:, 1] != after[:, 1]))))
    assert [float(r) for r in ts.reward] == list(o.rewards())
    for p, obs in enumerate(ts.observation):
      assert np.array_equal(obs["RGB"], o.render_agent(p))
    assert np.array_equal(ts.observation[0]["WORLD.RGB"], o.render_world())
  assert moved_and_turned   # some avatar changed cell and facing in ONE step
  env.close()


def test_lab2d_environment_takes_any_action_dict_dmlab2d_would():
  """dmlab2d.Environment.step(dict): independent "<player>.<field>" entries, absent
  keys keep their default, values are checked against the action spec."""
  from oracle import oracle as oracle_lib
  pk = engine.load_pack("commons_harvest__open")
  raw = lab2d_env.Environment("commons_harvest__open", ("default",) * 4,
                              engine=OracleEngine(pk, 21, 4))
  o = oracle_lib.Oracle(pk, 21, 4)
  o.reset()
  raw.reset()
  spec = raw.action_spec()
  assert (spec["1.move"].minimum, spec["1.move"].maximum) == (0, 4)
  assert (spec["3.turn"].minimum, spec["3.turn"].maximum) == (-1, 1)
  assert (spec["4.fireZap"].minimum, spec["4.fireZap"].maximum) == (0, 1)
  ts = raw.step({"1.move": 1, "1.turn": -1, "1.fireZap": 1, "3.turn": 1})
  o.step_fields(np.array([[1, -1, 1], [0, 0, 0], [0, 1, 0], [0, 0, 0]], np.int32))
  for p in range(4):
    assert np.array_equal(ts.observation[f"{p + 1}.RGB"], o.render_agent(p))
  with pytest.raises(ValueError):
    raw.step({"1.move": 5})
  with pytest.raises(ValueError):
    raw.step({"1.turn": 2})
  with pytest.raises(ValueError):
    raw.step({"9.move": 1})
  with pytest.raises(ValueError):
    raw.step({"1.fireClean": 1})    # not a field of this level
  raw.close()
