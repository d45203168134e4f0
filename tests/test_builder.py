"""`meltingpot_amd.builder.builder(lab2d_settings, prefab_overrides, env_seed)` — the
reference's L3 entry point (utils/substrates/builder.py:142-192) on this engine: any
settings dict of an implemented level, lowered at run time.  The fixture
(tests/golden/clean_up_modified_settings.pkl, written by
tests/tools/make_settings_fixture.py where the reference tree is) is the reference's
own clean_up settings with an EDITED map — no committed pack's — and the tests add
prefab overrides on AppleGrow's kwargs.

CPU: the semantics of the overrides, the lowering of the modified config, the
reference's unmodified wrapper stack on the returned environment (oracle-backed).
GPU: the HIP engine on the run-time pack against the oracle on the same pack."""
import os
import pickle

import numpy as np
import pytest

import util
from meltingpot_amd import builder, engine, pack as pack_lib, refshim

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                      "clean_up_modified_settings.pkl")


@pytest.fixture(scope="module")
def fixture():
  with open(GOLDEN, "rb") as f:
    return pickle.load(f)


def _oracle_engine(pack_bytes, seed, players):
  from oracle_engine import OracleEngine
  return OracleEngine(pack_bytes, seed, players)


def _builder_on_the_oracle(monkeypatch, *args, **kwargs):
  """`builder.builder(...)` with the engine it creates replaced by the CPU oracle behind the
  same interface (the injection lives HERE, in the tests: the product entry point has the
  reference's signature and creates a HIP engine, nothing else)."""
  def engine_class(pack_bytes, num_worlds, *, device=0, auto_reset=True, num_players=0,
                   base_seed=0, literal_seed=False):
    assert num_worlds == 1 and auto_reset and literal_seed
    return _oracle_engine(pack_bytes, base_seed, num_players)
  with monkeypatch.context() as m:
    m.setattr(builder.engine_lib, "Engine", engine_class)
    return builder.builder(*args, **kwargs)


def test_prefab_overrides_follow_the_reference(fixture):
  """builder.py:70-87: the first component of that name, its kwargs, in a COPY of the
  settings; an unknown prefab is a ValueError; `simulation.gameObjects` exists after."""
  settings = fixture["lab2d_settings"]
  before = pickle.dumps(settings)
  level, pack_bytes, cfg = builder.lower_settings(settings, fixture["prefab_overrides"])
  assert pickle.dumps(settings) == before, "the caller's settings were modified"
  assert level == "clean_up"
  t = pack_lib.loads(pack_bytes)
  stock = pack_lib.loads(engine.load_pack("clean_up"))
  # AppleGrow kwargs: maxAppleGrowthRate, thresholdDepletion, thresholdRestoration
  assert list(t["cu_f64"][:3]) == [0.5, 0.9, 0.0] and list(stock["cu_f64"][:3]) == [0.05, 0.4, 0.0]
  # ... and the edited map: one dirt site less, four more apple sites, one spawn point less
  assert len(t["dirt_cells"]) == len(stock["dirt_cells"]) - 1
  assert len(t["apple_cells"]) == len(stock["apple_cells"]) + 4
  assert len(t["spawn_cells"]) == len(stock["spawn_cells"]) - 1
  assert not np.array_equal(t["init_grid"], stock["init_grid"])
  # without overrides the kwargs are the config's
  _, plain_pack, _ = builder.lower_settings(settings)
  assert list(pack_lib.loads(plain_pack)["cu_f64"][:3]) == [0.05, 0.4, 0.0]
  with pytest.raises(ValueError, match="not available in `prefabs`"):
    builder.lower_settings(settings, {"no_such_prefab": {"AppleGrow": {"x": 1}}})
  with pytest.raises(ValueError, match="No component with name"):
    builder.lower_settings(settings, {"potential_apple": {"NoSuchComponent": {"x": 1}}})
  bare = pickle.loads(before)
  del bare["simulation"]["gameObjects"]
  with pytest.raises(AssertionError):   # lower: fewer avatars than numPlayers
    builder.lower_settings(bare)
  assert cfg.individual_observation_names == ["RGB", "READY_TO_SHOOT",
                                              "NUM_OTHERS_WHO_CLEANED_THIS_STEP"]
  assert cfg.timestep_spec["WORLD.RGB"].shape == (168, 240, 3)


def test_a_level_without_an_engine_is_refused(fixture):
  settings = pickle.loads(pickle.dumps(fixture["lab2d_settings"]))
  settings["levelName"] = "boat_race"
  with pytest.raises(NotImplementedError):
    builder.lower_settings(settings)
  settings = pickle.loads(pickle.dumps(fixture["lab2d_settings"]))
  settings["simulation"]["scene"]["components"].append({"component": "Inventor", "kwargs": {}})
  with pytest.raises(NotImplementedError, match="Inventor"):
    builder.lower_settings(settings)


def test_avatar_prefab_path(fixture):
  """builder.py:90-130: an 'avatar' prefab + playerPalettes builds the avatar objects
  (game_object_utils.py:85-135); buildAvatars defers to Lua, which this engine has not."""
  settings = pickle.loads(pickle.dumps(fixture["lab2d_settings"]))
  sim = settings["simulation"]
  avatars = [o for o in sim["gameObjects"]
             if any(c["component"] == "Avatar" for c in o["components"])]
  proto = pickle.loads(pickle.dumps(avatars[0]))
  app = builder._first_named_component(proto, "Appearance")["kwargs"]
  base = app["spriteNames"][0].rstrip("0123456789")
  for sc in builder._first_named_component(proto, "StateManager")["kwargs"]["stateConfigs"]:
    if sc.get("sprite") == app["spriteNames"][0]:
      sc["sprite"] = base
  app["spriteNames"][0] = base
  palettes = [builder._first_named_component(a, "Appearance")["kwargs"]["palettes"][0]
              for a in avatars]
  sim["gameObjects"] = [o for o in sim["gameObjects"] if o not in avatars]
  sim["prefabs"]["avatar"] = proto
  with pytest.raises(NotImplementedError, match="playerPalettes"):
    builder.lower_settings(settings)
  sim["playerPalettes"] = palettes
  _, rebuilt, _ = builder.lower_settings(settings)
  _, original, _ = builder.lower_settings(fixture["lab2d_settings"])
  # (state and sprite ids differ — the prefab table has one more entry — the worlds do not:
  # same draws, same pixels, same rewards)
  wa, wb = _oracle_engine(rebuilt, 3, 7), _oracle_engine(original, 3, 7)
  wa.reset(); wb.reset()
  rng = np.random.default_rng(1)
  for _ in range(30):
    f = rng.integers(0, 2, size=(7, 4)).astype(np.int32)
    wa.step_fields(f); wb.step_fields(f)
  from meltingpot_amd import engine as E
  # (not the per-agent view: clean_up's avatars carry a per-player spriteMap — own sprite
  # -> "Self" — which build_avatar_objects, here as in the reference, copies from the prefab)
  for kind in (E.OBS_WORLD_RGB, E.OBS_REWARD, E.OBS_READY_TO_SHOOT):
    assert np.array_equal(wa.observe_host(kind), wb.observe_host(kind)), kind
  sim["buildAvatars"] = True
  with pytest.raises(NotImplementedError, match="buildAvatars"):
    builder.lower_settings(settings)


def test_environment_on_a_runtime_pack_oracle_backed(fixture, monkeypatch):
  """The returned object is the dmlab2d.Environment duck type: flat "N.KEY"
  observations with the run-time config's specs, raw field actions."""
  env = _builder_on_the_oracle(monkeypatch, fixture["lab2d_settings"],
                               fixture["prefab_overrides"], env_seed=5)
  spec = env.observation_spec()
  assert spec["3.RGB"].shape == (88, 88, 3) and spec["WORLD.RGB"].shape == (168, 240, 3)
  ts = env.reset()
  assert ts.step_type == 0 and ts.reward is None
  rng = np.random.default_rng(0)
  total = 0.0
  for _ in range(120):
    ts = env.step({f"{p + 1}.move": int(rng.integers(0, 5)) for p in range(7)})
    for name, s in spec.items():
      got = np.asarray(ts.observation[name])
      assert got.shape == s.shape and got.dtype == s.dtype, name
    total += sum(float(ts.observation[f"{p + 1}.REWARD"]) for p in range(7))
  assert total > 0, "apples grow at rate 0.5 under the override: somebody eats within 120 steps"
  env.close()


@pytest.mark.skipif(not os.path.isdir(refshim.DEFAULT_REFERENCE_ROOT),
                    reason="reference tree not present (GPU box)")
def test_reference_build_substrate_stack_on_the_returned_environment(fixture, monkeypatch):
  """utils/substrates/substrate.py:107-139 `build_substrate`, line by line, with
  `builder.builder` replaced by this package's — the reference's own wrappers and its
  own conformance check (testing/substrates.py:22-68), unmodified, on a modified
  config; and the fixture really is what the reference's build() makes, map aside."""
  ref = refshim.load_reference_wrappers()
  settings, module, config = refshim.build_settings("clean_up", ("default",) * 7)
  ours = pickle.loads(pickle.dumps(fixture["lab2d_settings"]))
  theirs = builder._plain(settings)
  ours["simulation"].pop("map"); theirs["simulation"].pop("map")
  assert ours == theirs
  env = _builder_on_the_oracle(monkeypatch, fixture["lab2d_settings"],
                               fixture["prefab_overrides"], env_seed=11)
  env = ref.observables_wrapper.ObservablesWrapper(env)
  env = ref.multiplayer_wrapper.Wrapper(
      env, individual_observation_names=config.individual_observation_names,
      global_observation_names=config.global_observation_names)
  env = ref.discrete_action_wrapper.Wrapper(env, action_table=config.action_set)
  env = ref.collective_reward_wrapper.CollectiveRewardWrapper(env)
  env = ref.substrate.Substrate(env)
  case = ref.testing_substrates.SubstrateTestCase()
  with env:
    case.assert_step_matches_specs(env)
    assert len(env.action_spec()) == 7
    assert env.action_spec()[0].num_values == len(config.action_set)


# ---------------------------------------------------------------------------- GPU


@pytest.mark.gpu
def test_modified_config_on_the_hip_engine(fixture):
  """A runtime-lowered, modified clean_up on the GPU: 24 worlds x 200 steps, the
  fused launch drawing both views, state / scalars / pixels against the oracle on
  the same pack (apples grow fast under the override: eating, rewards and the
  NUM_OTHERS cumulants are exercised, not just movement)."""
  import test_gpu_parity as T
  from meltingpot_amd import substrate
  _, pack_bytes, _ = builder.lower_settings(
      fixture["lab2d_settings"], fixture["prefab_overrides"],
      action_set=substrate.get_config("clean_up").action_set)
  T._run(pack_bytes, n=24, steps=200, seed=77, rgb_every=40, fused="both")
  T._run(pack_bytes, n=9, steps=60, seed=78, weights=[1, 2, 2, 2, 2, 1, 1, 3, 6],
         rgb_every=20, fused="world")


@pytest.mark.gpu
def test_builder_environment_on_the_hip_engine(fixture, monkeypatch):
  """`builder.builder(...)` itself on the GPU: the flat timesteps of 200 steps equal
  those of the same class over the oracle, leaf by leaf, events included."""
  rng = np.random.default_rng(9)
  hip = builder.builder(fixture["lab2d_settings"], fixture["prefab_overrides"], env_seed=21)
  cpu = _builder_on_the_oracle(monkeypatch, fixture["lab2d_settings"],
                               fixture["prefab_overrides"], env_seed=21)
  a, b = hip.reset(), cpu.reset()
  for step in range(200):
    assert a.step_type == b.step_type and a.discount == b.discount, step
    assert sorted(a.observation) == sorted(b.observation)
    for k in a.observation:
      assert np.array_equal(a.observation[k], b.observation[k]), (step, k)
    assert hip.events() == cpu.events() or str(hip.events()) == str(cpu.events()), step
    act = {}
    for p in range(7):
      act[f"{p + 1}.move"] = int(rng.integers(0, 5))
      act[f"{p + 1}.turn"] = int(rng.integers(-1, 2))
      act[f"{p + 1}.fireZap"] = int(rng.random() < 0.15)
      act[f"{p + 1}.fireClean"] = int(rng.random() < 0.3)
    a, b = hip.step(act), cpu.step(act)
  hip.close(); cpu.close()
