"""The host-side mirror of the reference's substrate API
(meltingpot/substrate.py, utils/substrates/substrate.py)."""
import numpy as np
import pytest

from meltingpot_amd import pack, substrate


def test_registry_and_config_match_the_reference_config():
  assert "clean_up" in substrate.SUBSTRATES
  cfg = substrate.get_config("clean_up")
  # clean_up.py:806-838
  assert cfg.individual_observation_names == [
      "RGB", "READY_TO_SHOOT", "NUM_OTHERS_WHO_CLEANED_THIS_STEP"]
  assert cfg.global_observation_names == ["WORLD.RGB"]
  assert cfg.action_spec.num_values == 9 and cfg.action_spec.dtype == np.int64
  assert cfg.valid_roles == {"default"} and len(cfg.default_player_roles) == 7
  assert cfg.timestep_spec["RGB"].shape == (88, 88, 3)
  assert cfg.timestep_spec["WORLD.RGB"].shape == (168, 240, 3)
  with pytest.raises(ValueError):
    substrate.get_config("no_such_substrate")


def test_action_set_is_the_table_the_engine_indexes(clean_up_pack):
  cfg = substrate.get_config("clean_up")
  tab = pack.loads(clean_up_pack)["action_table"].reshape(-1, 4)
  names = ("move", "turn", "fireZap", "fireClean")
  assert len(cfg.action_set) == len(tab)
  for row, act in zip(tab, cfg.action_set):
    assert tuple(int(x) for x in row) == tuple(act[n] for n in names)


def test_commons_config_and_action_set(commons_pack):
  cfg = substrate.get_config("commons_harvest__open")
  assert cfg.individual_observation_names == ["RGB", "READY_TO_SHOOT"]
  assert cfg.timestep_spec["WORLD.RGB"].shape == (144, 192, 3)  # :555
  tab = pack.loads(commons_pack)["action_table"].reshape(-1, 4)
  assert len(cfg.action_set) == len(tab) == 8
  for row, act in zip(tab, cfg.action_set):
    assert tuple(int(x) for x in row[:3]) == (act["move"], act["turn"], act["fireZap"])


def test_territory_config_and_action_set(territory_pack):
  cfg = substrate.get_config("territory__rooms")
  assert cfg.individual_observation_names == ["RGB", "READY_TO_SHOOT"]   # territory.py:846-849
  assert cfg.timestep_spec["WORLD.RGB"].shape == (168, 168, 3)           # territory__rooms.py:98
  assert len(cfg.default_player_roles) == 9                               # :102
  tab = pack.loads(territory_pack)["action_table"].reshape(-1, 4)
  assert len(cfg.action_set) == len(tab) == 9                             # territory.py:592-602
  for row, act in zip(tab, cfg.action_set):
    assert tuple(int(x) for x in row) == (
        act["move"], act["turn"], act["fireZap"], act["fireClaim"])


def test_invalid_roles_raise_value_error_like_the_reference():
  # configs/substrates/__init__.py:42-45 — checked before any device is touched
  with pytest.raises(ValueError, match="Invalid roles"):
    substrate.build("clean_up", roles=("default",) * 6 + ("villain",))


def test_spec_classes_validate_like_dm_env():
  a = substrate.Array((2,), np.float64, "x")
  a.validate(np.zeros(2))
  with pytest.raises(ValueError):
    a.validate(np.zeros(3))
  with pytest.raises(ValueError):
    a.validate(np.zeros(2, np.float32))
  d = substrate.DiscreteArray(9)
  assert d.maximum == 8 and d.minimum == 0
  with pytest.raises(ValueError):
    d.validate(np.int64(9))
  assert substrate.StepType.LAST.last() and not substrate.StepType.MID.first()
  # equality is dm_env's: shape and dtype (bounded: and the bounds), never the name
  assert substrate.Array((2,), np.float64, "x") == substrate.Array((2,), np.float64, "1.x")
  assert substrate.Array((2,), np.float64) != substrate.Array((2,), np.float32)
  assert substrate.DiscreteArray(9, name="action") == substrate.DiscreteArray(9, name="1.action")
  assert substrate.DiscreteArray(9) != substrate.DiscreteArray(8)
  assert substrate.DiscreteArray(9) != substrate.Array((), np.int64)       # BoundedArray.__eq__ wants bounds


@pytest.mark.gpu
def test_step_matches_specs():
  """meltingpot/testing/substrates.py:22-68 (assert_step_matches_specs), the
  check the reference runs on every substrate (substrate_test.py:24-47)."""
  cfg = substrate.get_config("clean_up")
  with substrate.build("clean_up", roles=cfg.default_player_roles) as env:
    first = env.reset()
    assert first.first() and first.discount == 0.0
    action = [int(spec.maximum) for spec in env.action_spec()]
    timestep = env.step(action)
    env.discount_spec().validate(np.float64(timestep.discount))
    reward_spec = env.reward_spec()
    assert len(reward_spec) == len(timestep.reward) == 7
    for n, spec in enumerate(reward_spec):
      spec.validate(timestep.reward[n])
    observation_specs = env.observation_spec()
    assert len(observation_specs) == len(timestep.observation)
    for observation, spec in zip(timestep.observation, observation_specs):
      assert set(spec) == set(observation)
      for key in spec:
        spec[key].validate(observation[key])
    with pytest.raises(ValueError):
      env.step([0] * 6)
    with pytest.raises(ValueError):
      env.step([9] + [0] * 6)


@pytest.mark.gpu
def test_episode_loop_and_batched_leaves():
  """The caller loop of utils/evaluation/evaluation.py:37-49 on one world, and
  the batched form: every leaf is a device tensor with a leading [N] axis."""
  import torch
  cfg = substrate.get_config("clean_up")
  env = substrate.build("clean_up", roles=cfg.default_player_roles)
  timestep = env.reset()
  rng = np.random.default_rng(0)
  steps = 0
  while not timestep.last() and steps < 50:
    timestep = env.step(rng.integers(0, 9, 7))
    steps += 1
  assert steps == 50 and timestep.mid()
  env.close()

  n = 32
  env = substrate.build("clean_up", roles=cfg.default_player_roles, num_worlds=n)
  ts = env.reset()
  assert ts.observation["RGB"].shape == (n, 7, 88, 88, 3) and ts.observation["RGB"].is_cuda
  assert ts.observation["WORLD.RGB"].shape == (n, 168, 240, 3)
  assert ts.reward.shape == (n, 7) and ts.reward.dtype == torch.float64
  assert (ts.step_type == 0).all()
  acts = torch.randint(0, 9, (n, 7), device=ts.reward.device, dtype=torch.int32)
  ts = env.step(acts)
  assert (ts.step_type == 1).all() and (ts.discount == 1.0).all()
  assert ts.observation["COLLECTIVE_REWARD"].shape == (n,)
  env.close()


def test_subject_is_the_reactivex_slice_the_reference_uses():
  seen, done = [], []
  sub = substrate.Subject()
  d = sub.subscribe(on_next=seen.append, on_completed=lambda: done.append(1))
  sub.on_next(1); sub.on_next(2)
  d.dispose()
  sub.on_next(3)
  sub.subscribe(on_next=seen.append, on_completed=lambda: done.append(2))
  sub.on_completed()
  assert seen == [1, 2] and done == [2]


@pytest.mark.gpu
def test_observables_emit_actions_timesteps_and_events():
  """substrate.py:56-104: reset/step push onto the action / timestep / events
  subjects, close completes them."""
  cfg = substrate.get_config("clean_up")
  actions, timesteps, events, completed = [], [], [], []
  env = substrate.build("clean_up", roles=cfg.default_player_roles)
  obs = env.observables()
  obs.action.subscribe(on_next=actions.append)
  obs.timestep.subscribe(on_next=timesteps.append, on_completed=lambda: completed.append(1))
  obs.events.subscribe(on_next=events.append)
  env.reset()
  for _ in range(3):
    env.step([8] * 7)   # FIRE_CLEAN
  env.close()
  assert len(actions) == 3 and len(timesteps) == 4 and completed == [1]
  assert timesteps[0].step_type == substrate.StepType.FIRST
  assert [name for name, _ in events[:7]] == ["AvatarStarted"] * 7
  assert all(isinstance(payload, dict) for _, payload in events)
  # a batch: `events` stays reference-shaped (world 0), `events_batched` carries
  # every world's events tagged with the world
  env = substrate.build("clean_up", roles=cfg.default_player_roles, num_worlds=3, env_seed=9)
  seen, first = [], []
  env.observables().events_batched.subscribe(on_next=seen.append)
  env.observables().events.subscribe(on_next=first.append)
  env.reset()
  env.close()
  assert sorted(w for w, _ in seen) == [0] * 7 + [1] * 7 + [2] * 7
  assert all(name == "AvatarStarted" for _, (name, _) in seen)
  assert first == [e for w, e in seen if w == 0]


@pytest.mark.gpu
def test_commons_step_matches_specs():
  cfg = substrate.get_config("commons_harvest__open")
  with substrate.build("commons_harvest__open", roles=cfg.default_player_roles) as env:
    env.reset()
    timestep = env.step([int(spec.maximum) for spec in env.action_spec()])
    assert len(timestep.reward) == len(cfg.default_player_roles) == 7   # commons_harvest__open.py:560
    for observation, spec in zip(timestep.observation, env.observation_spec()):
      assert set(spec) == set(observation)
      for key in spec:
        spec[key].validate(observation[key])


@pytest.mark.gpu
def test_territory_step_matches_specs():
  cfg = substrate.get_config("territory__rooms")
  with substrate.build("territory__rooms", roles=cfg.default_player_roles) as env:
    env.reset()
    timestep = env.step([int(spec.maximum) for spec in env.action_spec()])
    assert len(timestep.reward) == 9
    for observation, spec in zip(timestep.observation, env.observation_spec()):
      assert set(spec) == set(observation)
      for key in spec:
        spec[key].validate(observation[key])


def test_coins_config_and_action_set(coins_pack):
  cfg = substrate.get_config("coins")
  # coins.py:467-486
  assert cfg.individual_observation_names == ["RGB", "MISMATCHED_COIN_COLLECTED_BY_PARTNER"]
  assert cfg.timestep_spec["WORLD.RGB"].shape == (136, 136, 3)
  tab = pack.loads(coins_pack)["action_table"].reshape(-1, 4)
  assert len(cfg.action_set) == len(tab) == 7                    # coins.py:442-450
  for row, act in zip(tab, cfg.action_set):
    assert tuple(int(x) for x in row[:2]) == (act["move"], act["turn"])


@pytest.mark.gpu
def test_coins_step_matches_specs():
  cfg = substrate.get_config("coins")
  with substrate.build("coins", roles=cfg.default_player_roles) as env:
    env.reset()
    timestep = env.step([1, 1])
    assert len(timestep.reward) == 2
    for observation, spec in zip(timestep.observation, env.observation_spec()):
      assert set(spec) == set(observation)
      for key in spec:
        spec[key].validate(observation[key])


@pytest.mark.gpu
@pytest.mark.parametrize("name,players,nact", [("clean_up", 7, 9), ("clean_up", 3, 9),
                                               ("commons_harvest__open", 16, 8)])
def test_flat_lab2d_environment_on_the_hip_engine(name, players, nact):
  """`lab2d_env.Environment` is the `dmlab2d.Environment` duck type under the
  reference's wrapper stack.  tests/test_reference_wrappers.py runs the
  reference's unmodified wrappers on it with the CPU oracle as the world (the
  reference tree is not on the GPU box); here the same class runs on the HIP
  engine and must hand the wrappers the very same flat timesteps, events
  included."""
  from meltingpot_amd import engine, lab2d_env
  from oracle_engine import OracleEngine
  roles = ("default",) * players
  seed = 4242
  gpu = lab2d_env.Environment(name, roles, env_seed=seed)
  cpu = lab2d_env.Environment(name, roles,
                              engine=OracleEngine(engine.load_pack(name), seed, players))
  assert gpu.action_spec() == cpu.action_spec()
  assert gpu.observation_spec() == cpu.observation_spec()
  cfg = substrate.get_config(name)
  a, b = gpu.reset(), cpu.reset()
  rng = np.random.default_rng(0)
  for _ in range(40):
    assert a.step_type == b.step_type and a.discount == b.discount and a.reward is b.reward is None
    assert set(a.observation) == set(b.observation)
    for k in a.observation:
      assert np.array_equal(a.observation[k], b.observation[k]), k
    assert sorted(map(repr, gpu.events())) == sorted(map(repr, cpu.events()))
    flat = {}
    for p, i in enumerate(rng.integers(0, nact, players)):
      for key, value in cfg.action_set[i].items():
        flat[f"{p + 1}.{key}"] = value
    a, b = gpu.step(flat), cpu.step(flat)
  # not a row of ACTION_SET, but an action dmlab2d takes: forward while turning
  a, b = gpu.step({"1.move": 1, "1.turn": 1}), cpu.step({"1.move": 1, "1.turn": 1})
  for k in a.observation:
    assert np.array_equal(a.observation[k], b.observation[k]), k
  with pytest.raises(ValueError):
    gpu.step({"1.move": 9})             # outside the action spec
  gpu.close(); cpu.close()


@pytest.mark.gpu
def test_env_seed_semantics_of_the_reference_builder():
  """utils/substrates/builder_test.py:47-75 on the HIP engine: the same env_seed —
  0 included, it is a seed like any other (builder.py:174-181) — gives the same
  WORLD.RGB at every reset of two separately built substrates; consecutive
  episodes of one substrate differ; unseeded substrates differ from each other;
  and world w of a batch built with env_seed s is the single world built with
  env_seed s + w."""
  roles = ("default",) * 4
  for seed in (42, 0, 12481632, -5):
    with substrate.build("commons_harvest__open", roles=roles, env_seed=seed) as a, \
         substrate.build("commons_harvest__open", roles=roles, env_seed=seed) as b:
      last = None
      for episode in range(4):
        oa = a.reset().observation[0]["WORLD.RGB"]
        ob = b.reset().observation[0]["WORLD.RGB"]
        assert np.array_equal(oa, ob), (seed, episode)
        assert last is None or not np.array_equal(last, oa), (seed, episode)
        last = oa
  with substrate.build("commons_harvest__open", roles=roles) as a, \
       substrate.build("commons_harvest__open", roles=roles) as b:
    assert not np.array_equal(a.reset().observation[0]["WORLD.RGB"],
                              b.reset().observation[0]["WORLD.RGB"])
  with substrate.build("commons_harvest__open", roles=roles, env_seed=0, num_worlds=3) as batch, \
       substrate.build("commons_harvest__open", roles=roles, env_seed=2) as single:
    batched = batch.reset().observation["WORLD.RGB"][2].cpu().numpy()
    assert np.array_equal(batched, single.reset().observation[0]["WORLD.RGB"])


@pytest.mark.gpu
def test_debug_observations_through_build_substrate():
  """`build_substrate(individual_observations=[..., "PLAYER_CLEANED", "POSITION"])`: the debug
  observations a reference config built with _ENABLE_DEBUG_OBSERVATIONS reports
  (clean_up.py:751-784) as leaves of the batched timestep, against the oracle."""
  import pickle
  import os
  import torch
  from oracle import oracle as oracle_lib
  from meltingpot_amd import builder
  here = os.path.dirname(os.path.abspath(__file__))
  with open(os.path.join(here, "golden", "clean_up_modified_settings.pkl"), "rb") as f:
    settings = pickle.load(f)["lab2d_settings"]
  cfg = substrate.get_config("clean_up")
  names = ["RGB", "READY_TO_SHOOT", "PLAYER_CLEANED", "PLAYER_ATE_APPLE",
           "NUM_OTHERS_WHO_ATE_THIS_STEP", "POSITION", "ORIENTATION"]
  env = substrate.build_substrate(lab2d_settings=settings, individual_observations=names,
                                  global_observations=["WORLD.RGB"], action_table=cfg.action_set,
                                  num_worlds=6, env_seed=70)
  _, pack_bytes, _ = builder.lower_settings(settings, action_set=cfg.action_set)
  refs = [oracle_lib.Oracle(pack_bytes, 70 + w, 7) for w in range(6)]
  ts = env.reset()
  for o in refs:
    o.reset()
  assert set(ts.observation) == set(names) | {"WORLD.RGB", "COLLECTIVE_REWARD"}
  assert env.observation_spec()[0]["PLAYER_CLEANED"].dtype == np.float64
  rng = np.random.default_rng(5)
  cleaned = 0.0
  for _ in range(40):
    a = rng.choice([1, 2, 3, 4, 7, 8, 8], size=(6, 7)).astype(np.int32)
    ts = env.step(torch.from_numpy(a).to(env.engine.device))
    for w, o in enumerate(refs):
      o.step(a[w])
      m = o.debug_metrics()
      assert np.array_equal(ts.observation["PLAYER_CLEANED"][w].cpu().numpy(), m[0]), w
      assert np.array_equal(ts.observation["PLAYER_ATE_APPLE"][w].cpu().numpy(), m[1]), w
      assert np.array_equal(ts.observation["NUM_OTHERS_WHO_ATE_THIS_STEP"][w].cpu().numpy(), m[3]), w
      _, avat, _ = o.dump()
      assert np.array_equal(ts.observation["POSITION"][w].cpu().numpy(), avat[:, :2]), w
      assert np.array_equal(ts.observation["ORIENTATION"][w].cpu().numpy(), avat[:, 2]), w
      cleaned += m[0].sum()
  assert cleaned > 0
  env.close()


def test_packed_event_rows_decode_to_the_reference_keys():
  """The events whose payload is packed into two ints come back under the REFERENCE's keys:
  extraction_pair (coop_mining/components.lua:220: player_a, player_b, ore_type) and gift
  (gift_refinements/components.lua:174-181: both indices, both roles, source_type,
  received_amount — the roles from the pack's `agent_roles`)."""
  from meltingpot_amd import engine as E
  rows = np.zeros((3, 4), np.int64)
  rows[0, 0] = 2
  rows[1] = (15, 3, (5 << 2) | 2, 0)
  rows[2] = (16, 2 | (1 << 4), 4 | (5 << 4), 0)
  roles = E.pack_agent_roles(E.load_pack("gift_refinements"))
  assert roles and set(roles) == {"none"}           # gift_refinements.py:350, one per avatar lowered
  assert E.pack_agent_roles(E.load_pack("clean_up")) == ()
  ev = E.Engine._decode_events(rows, 0, None, roles)
  assert ev == [
      ("extraction_pair", {"player_a": 3, "player_b": 5, "ore_type": 2}),
      ("gift", {"gifter_index": 2, "gifter_role": "none", "receipient_index": 4,
                "receipient_role": "none", "source_type": 1, "received_amount": 5})]
  for t, (name, keys) in E.EVENT_TYPES.items():
    if t in (15, 16):
      assert set(ev[t - 15][1]) == set(keys)
