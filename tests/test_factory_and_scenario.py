"""The factory surface of `meltingpot_amd.substrate` (meltingpot/substrate.py:57-113,
utils/substrates/substrate_factory.py:24-95, utils/substrates/substrate.py:107-139) and what
it is FOR: the reference's UNMODIFIED `Scenario` / `ScenarioFactory` / `Population` and the
episode loop of `utils/evaluation/evaluation.py:37-49`, imported from /root/reference by
`refshim.load_reference_scenarios`, standing on this package's `Substrate`.

CPU: the product's Substrate runs on `oracle_engine.OracleBatchEngine` (the test suite's
stand-in for `engine.Engine`: N CPU oracles behind the same interface, monkeypatched in).
GPU: the same code on the HIP engine; the focal timesteps of the two are compared leaf by
leaf (`tests/golden/scenario_*.npz` are the oracle-backed runs: the GPU box has no
reference tree, so the GPU test re-implements nothing of the reference and compares the
SUBSTRATE timesteps — full, unpartitioned — that the recorded scenario run produced)."""
import os
import pickle
import random

import numpy as np
import pytest

import util  # noqa: F401
from meltingpot_amd import builder, engine, pack as pack_lib, substrate

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HAVE_REFERENCE = os.path.isdir("/root/reference/meltingpot")
needs_reference = pytest.mark.skipif(not HAVE_REFERENCE, reason="no reference tree on this box")


@pytest.fixture
def oracle_backed(monkeypatch):
  """The product's Substrate on the CPU oracle (test infrastructure, injected HERE — the
  product has no engine= hook)."""
  from oracle_engine import OracleBatchEngine
  monkeypatch.setattr(substrate.engine_lib, "Engine", OracleBatchEngine)
  return OracleBatchEngine


@pytest.fixture(scope="module")
def modified_settings():
  with open(os.path.join(GOLDEN, "clean_up_modified_settings.pkl"), "rb") as f:
    return pickle.load(f)


# --------------------------------------------------------------------------
# the factory surface


def test_factory_methods_follow_the_reference():
  """substrate_factory.py:72-86 on a committed substrate."""
  f = substrate.get_factory("clean_up")
  assert isinstance(f, substrate.SubstrateFactory)
  assert f.valid_roles() == frozenset({"default"})
  assert f.default_player_roles() == ("default",) * 7
  spec = f.timestep_spec()
  assert isinstance(spec, substrate.TimeStep)                     # specs.py:149-166
  assert set(spec.observation) == {"RGB", "READY_TO_SHOOT", "NUM_OTHERS_WHO_CLEANED_THIS_STEP",
                                   "WORLD.RGB"}
  assert spec.observation["RGB"].shape == (88, 88, 3) and spec.observation["RGB"].name == "RGB"
  assert spec.step_type.dtype == np.int64 and spec.discount.maximum == 1
  assert f.action_spec().num_values == 9 and f.action_spec().dtype == np.int64
  with pytest.raises(ValueError, match="not in"):
    substrate.get_factory("boat_race__eight_races")
  with pytest.raises(TypeError):
    substrate.get_factory_from_config({"name": "clean_up"})


@needs_reference
@pytest.mark.parametrize("name", ["clean_up", "commons_harvest__open", "territory__rooms", "coins",
                                  "bach_or_stravinsky_in_the_matrix__repeated"])
def test_factory_agrees_with_the_reference_config(name):
  """get_factory(name) of this package against the reference's own config module:
  roles, per-player timestep spec, action spec."""
  from meltingpot_amd import refshim
  ref = refshim.load_config_module(name).get_config()
  f = substrate.get_factory(name)
  assert f.valid_roles() == frozenset(ref.valid_roles)
  assert len(f.default_player_roles()) == len(ref.default_player_roles)
  ref_obs = ref.timestep_spec if isinstance(ref.timestep_spec, dict) else ref.timestep_spec.observation
  ours = f.timestep_spec().observation
  assert set(ours) == set(ref_obs)
  for k, sp in ref_obs.items():
    assert tuple(ours[k].shape) == tuple(sp.shape) and np.dtype(ours[k].dtype) == np.dtype(sp.dtype), k
  assert f.action_spec().num_values == len(ref.action_set)


def test_an_edited_packed_config_is_honoured_or_refused(oracle_backed):
  """build_from_config does what the config SAYS: an edited action_set runs as that action
  table, edited observation lists pick the leaves, and what the committed pack cannot do — a
  different map size — raises instead of running the stock substrate silently."""
  cfg = substrate.get_config("clean_up")
  with cfg.unlocked():
    cfg.action_set = ({"move": 0, "turn": 0, "fireZap": 0, "fireClean": 0},
                      {"move": 1, "turn": 1, "fireZap": 0, "fireClean": 1})   # walk + turn + clean
    cfg.individual_observation_names = ["RGB"]
    cfg.global_observation_names = []
  env = substrate.build_from_config(cfg, roles=("default",) * 3, env_seed=11)
  try:
    assert [s.num_values for s in env.action_spec()] == [2, 2, 2]
    ts = env.reset()
    assert set(ts.observation[0]) == {"RGB", "COLLECTIVE_REWARD"}
    assert set(env.observation_spec()[0]) >= {"RGB", "COLLECTIVE_REWARD"}
    # ... and action 1 really is "forward + turn right + clean": the oracle on raw fields
    from oracle import oracle as oracle_lib
    o = oracle_lib.Oracle(engine.load_pack("clean_up"), 11, 3)
    o.reset()
    for _ in range(6):
      ts = env.step([1, 0, 1])
      o.step_fields(np.array([[1, 1, 0, 1], [0, 0, 0, 0], [1, 1, 0, 1]], np.int32))
      for p in range(3):
        assert np.array_equal(ts.observation[p]["RGB"], o.render_agent(p))
    with pytest.raises(ValueError):
      env.step([2, 0, 0])
  finally:
    env.close()
  bad = substrate.get_config("clean_up")
  bad.timestep_spec["WORLD.RGB"] = substrate.Array((176, 240, 3), np.uint8, "WORLD.RGB")
  with pytest.raises(ValueError, match="renders"):
    substrate.build_from_config(bad, roles=("default",) * 3)
  many = substrate.get_config("clean_up")
  with pytest.raises(ValueError, match="lowered for at most"):
    substrate.build_from_config(many, roles=("default",) * 40)
  odd = substrate.get_config("clean_up")
  odd.individual_observation_names = ["RGB", "HUNGER"]
  with pytest.raises(ValueError, match="HUNGER"):
    substrate.build_from_config(odd, roles=("default",) * 3)


def test_build_substrate_runs_the_settings_it_is_given(oracle_backed, modified_settings):
  """utils/substrates/substrate.py:107-139 batched: run-time settings (an EDITED clean_up
  map) at N worlds in one call; world w == the oracle on the run-time pack, seed env_seed + w."""
  settings = modified_settings["lab2d_settings"]
  cfg = substrate.get_config("clean_up")
  env = substrate.build_substrate(
      lab2d_settings=settings, individual_observations=["RGB", "READY_TO_SHOOT", "POSITION"],
      global_observations=["WORLD.RGB"], action_table=cfg.action_set, num_worlds=3, env_seed=40)
  try:
    assert env.num_worlds == 3 and env.num_players == 7
    _, pack_bytes, _ = builder.lower_settings(settings, action_set=cfg.action_set)
    assert env.engine.pack_bytes == pack_bytes != engine.load_pack("clean_up")
    from oracle import oracle as oracle_lib
    refs = [oracle_lib.Oracle(pack_bytes, 40 + w, 7) for w in range(3)]
    ts = env.reset()
    for o in refs:
      o.reset()
    assert set(ts.observation) == {"RGB", "READY_TO_SHOOT", "POSITION", "WORLD.RGB",
                                   "COLLECTIVE_REWARD"}
    assert env.observation_spec()[0]["POSITION"].shape == (2,)
    rng = np.random.default_rng(0)
    for _ in range(8):
      a = rng.integers(0, 9, size=(3, 7)).astype(np.int32)
      ts = env.step(a)
      for w, o in enumerate(refs):
        o.step(a[w])
        assert np.array_equal(ts.observation["WORLD.RGB"][w].numpy(), o.render_world())
        assert np.array_equal(ts.reward[w].numpy(), o.rewards())
  finally:
    env.close()
  with pytest.raises(ValueError, match="does not match action_spec"):
    substrate.build_substrate(lab2d_settings=settings, individual_observations=["RGB"],
                              global_observations=[], action_table=[{"move": 7, "turn": 0,
                                                                     "fireZap": 0, "fireClean": 0}])


@needs_reference
def test_a_modified_reference_config_runs_the_modified_map(oracle_backed):
  """meltingpot/substrate.py:75-113 with the REFERENCE's config object: get the config, edit
  what it builds (here: the ASCII map its module holds and AppleGrow's rate through the
  settings builder), build_from_config — the substrate that runs is the edited one."""
  from meltingpot_amd import refshim
  mod = refshim.load_config_module("clean_up")
  config = mod.get_config()
  stock_map = mod.ASCII_MAP

  def settings_builder(*, roles, config):
    settings = mod.build(roles, config)
    rows = settings["simulation"]["map"].strip("\n").split("\n")
    assert rows[8][1:5] == "    "
    rows[8] = rows[8][:1] + "BBBB" + rows[8][5:]   # four more potential apples, on the sand
    settings["simulation"]["map"] = "\n" + "\n".join(rows) + "\n"
    return settings

  with config.unlocked():
    config.lab2d_settings_builder = settings_builder
    config.action_spec = substrate.DiscreteArray(len(config.action_set))
    config.timestep_spec = substrate.timestep_spec_of({
        k: substrate.Array(v.shape, v.dtype, k) for k, v in dict(config.timestep_spec).items()})
  roles = config.default_player_roles
  env = substrate.build_from_config(config, roles=roles, env_seed=3)
  try:
    edited = builder.lower_settings(settings_builder(roles=roles, config=config),
                                    action_set=config.action_set)[1]
    assert env.engine.pack_bytes == edited
    t_edit, t_stock = pack_lib.loads(edited), pack_lib.loads(engine.load_pack("clean_up"))
    assert len(t_edit["apple_cells"]) == len(t_stock["apple_cells"]) + 4
    ts = env.reset()
    assert ts.observation[0]["WORLD.RGB"].shape == (168, 240, 3)
    from oracle import oracle as oracle_lib
    o = oracle_lib.Oracle(edited, 3, 7)
    o.reset()
    assert np.array_equal(ts.observation[0]["WORLD.RGB"], o.render_world())
  finally:
    env.close()
  assert mod.ASCII_MAP == stock_map


# --------------------------------------------------------------------------
# Scenario / evaluation on top


def _scripted_policy_class(ns):
  class Scripted(ns.policy.Policy):
    """Cycles through a fixed list of actions (state = position in the list)."""

    def __init__(self, actions):
      self._actions = tuple(actions)

    def initial_state(self):
      return 0

    def step(self, timestep, prev_state):
      assert set(timestep.observation) >= {"RGB", "READY_TO_SHOOT"}   # a bot sees everything
      return self._actions[prev_state % len(self._actions)], prev_state + 1

    def close(self):
      pass
  return Scripted


SCENARIO = dict(roles=("default",) * 7,
                is_focal=(True, False, True, True, False, False, True),
                bots_by_role={"default": ("cleaner", "walker")},
                permitted=frozenset({"RGB", "READY_TO_SHOOT", "COLLECTIVE_REWARD"}))


def _bots(ns):
  Scripted = _scripted_policy_class(ns)
  return {"cleaner": ns.fixed_action_policy.FixedActionPolicy(8),       # fireClean for ever
          "walker": Scripted([1, 1, 5, 1, 7, 6, 3])}


def _run_scenario(ns, env, steps, seed=5):
  """build_scenario (utils/scenarios/scenario.py:265-302) on `env`, `steps` steps of scripted
  focal actions; returns the focal timesteps and the full substrate timesteps seen on the way."""
  random.seed(seed)   # Population samples bot names with `random`
  scenario = ns.scenario.build_scenario(
      substrate=env, bots=_bots(ns), bots_by_role=SCENARIO["bots_by_role"],
      roles=SCENARIO["roles"], is_focal=SCENARIO["is_focal"],
      permitted_observations=SCENARIO["permitted"])
  full, joint = [], []
  env.observables().timestep.subscribe(on_next=full.append)
  env.observables().action.subscribe(on_next=joint.append)
  focal = [scenario.reset()]
  rng = np.random.default_rng(seed)
  for _ in range(steps):
    focal.append(scenario.step(tuple(int(a) for a in rng.integers(0, 9, size=4))))
  return scenario, focal, full, joint


PIXELS_AT = (0, 12, 25)   # timesteps whose pixels the recording keeps whole (the others: CRC-32)


def _record(full, actions):
  """The substrate's (full, unpartitioned) timesteps of a scenario run + the joint actions
  that produced them, as a flat dict of arrays: what the GPU suite replays on the HIP engine."""
  import zlib
  out = {"actions": np.asarray(actions, np.int32)}
  for t, ts in enumerate(full):
    out[f"{t}.step_type"] = np.asarray(int(ts.step_type))
    out[f"{t}.discount"] = np.asarray(ts.discount, np.float64)
    out[f"{t}.reward"] = np.asarray(ts.reward, np.float64)
    obs = ts.observation
    for k in obs[0]:
      if k == "WORLD.RGB":
        v = np.asarray(obs[0][k])[None]
      else:
        v = np.stack([np.asarray(o[k]) for o in obs])
      if v.dtype == np.uint8:
        out[f"{t}.{k}.crc"] = np.asarray([zlib.crc32(np.ascontiguousarray(x).tobytes()) for x in v],
                                         np.uint32)
        if t in PIXELS_AT:
          out[f"{t}.{k}"] = v
      else:
        out[f"{t}.{k}"] = v
  return out


@needs_reference
def test_reference_scenario_on_the_oracle_backed_substrate(oracle_backed):
  """The reference's Scenario, Population and policies, unmodified, on this package's
  Substrate (get_factory -> build): focal players see only their own, permitted
  observations; the background bots act; events() is empty; specs are partitioned."""
  from meltingpot_amd import refshim
  ns = refshim.load_reference_scenarios()
  env = substrate.get_factory("clean_up").build(SCENARIO["roles"], env_seed=77)
  scenario, focal, full, joint = _run_scenario(ns, env, 25)
  try:
    assert isinstance(scenario, ns.scenario.Scenario)
    assert len(scenario.action_spec()) == 4 and len(scenario.observation_spec()) == 4
    assert set(scenario.observation_spec()[0]) == set(SCENARIO["permitted"])
    assert len(scenario.reward_spec()) == 4
    assert scenario.events() == ()
    assert len(focal) == len(full) == 26
    focal_slots = [i for i, f in enumerate(SCENARIO["is_focal"]) if f]
    for ft, st in zip(focal, full):
      assert ft.step_type == st.step_type and ft.discount == st.discount
      assert len(ft.reward) == 4 and len(ft.observation) == 4
      for j, i in enumerate(focal_slots):
        assert ft.reward[j] == st.reward[i]
        assert set(ft.observation[j]) == set(SCENARIO["permitted"])
        assert np.array_equal(ft.observation[j]["RGB"], st.observation[i]["RGB"])
    # the background really played: player 2 (a bot) cleaned or walked, so it is not where
    # an idle avatar would be; the cleaner bot's beam shows up as a non-zero cumulant
    assert any(np.asarray(t.observation[0]["NUM_OTHERS_WHO_CLEANED_THIS_STEP"]) > 0 for t in full[1:])
    with pytest.raises(ValueError, match="focal actions"):
      scenario.step((0, 0, 0))
    assert len(scenario.observation()) == 4
  finally:
    scenario.close()
  # the recorded run the GPU suite compares the HIP engine with
  golden = os.path.join(GOLDEN, "scenario_clean_up_full_timesteps.npz")
  assert len(joint) == 25 and all(len(a) == 7 for a in joint)
  flat = _record(full, joint)
  if os.environ.get("MP_WRITE_GOLDEN"):
    np.savez_compressed(golden, **flat)
  with np.load(golden) as z:
    assert set(z.files) == set(flat)
    for k in z.files:
      assert np.array_equal(z[k], flat[k]), k


@needs_reference
def test_reference_scenario_factory_and_episode_loop(oracle_backed):
  """scenario_factory.py:29-130 takes a substrate FACTORY: this package's, as
  meltingpot/scenario.py:119-134 would hand it over; then the reference's episode loop
  (utils/evaluation/evaluation.py:37-49 run_episode) and its return bookkeeping
  (return_subject.py) on the built scenario."""
  from meltingpot_amd import refshim
  ns = refshim.load_reference_scenarios()
  bots = _bots(ns)
  factories = {name: ns.policy_factory.PolicyFactory(
      timestep_spec=substrate.get_factory("clean_up").timestep_spec(),
      action_spec=substrate.get_factory("clean_up").action_spec(),
      builder=(lambda p=p: p)) for name, p in bots.items()}
  # A SHORT episode, made the way a user of the reference makes one: the reference's own
  # config object with its lab2d settings edited (`maxEpisodeLengthFrames`), handed to
  # get_factory_from_config — which lowers exactly those settings at run time.
  mod = refshim.load_config_module("clean_up")
  cfg = mod.get_config()

  def short_settings(*, roles, config):
    settings = mod.build(roles, config)
    assert settings["maxEpisodeLengthFrames"] == 5000
    settings["maxEpisodeLengthFrames"] = 40
    return settings
  with cfg.unlocked():
    # (what configs/substrates/__init__.py:58-67 attaches to a config)
    cfg.lab2d_settings_builder = short_settings
    cfg.action_spec = substrate.DiscreteArray(len(cfg.action_set))
    cfg.timestep_spec = substrate.timestep_spec_of({
        k: substrate.Array(v.shape, v.dtype, k) for k, v in dict(cfg.timestep_spec).items()})
  sub_factory = substrate.get_factory_from_config(cfg)

  class Seeded:   # (the factory is the product's; its build() takes Substrate's keyword arguments)
    def __getattr__(self, name):
      return getattr(sub_factory, name)

    def build(self, roles):
      return sub_factory.build(roles, env_seed=5)
  sf = ns.scenario_factory.ScenarioFactory(
      substrate=Seeded(), bots=factories, bots_by_role=SCENARIO["bots_by_role"],
      roles=SCENARIO["roles"], is_focal=SCENARIO["is_focal"],
      permitted_observations=SCENARIO["permitted"])
  assert sf.num_focal_players() == 4
  assert set(sf.timestep_spec().observation) == {"RGB", "READY_TO_SHOOT"}
  assert sf.action_spec().num_values == 9
  random.seed(1)
  scenario = sf.build()
  try:
    Scripted = _scripted_policy_class(ns)
    focal_population = ns.population.Population(
        policies={"me": Scripted([1, 7, 8, 5])}, names_by_role={"default": ("me",)},
        roles=("default",) * 4)
    returns = []
    rs = ns.return_subject.ReturnSubject()
    focal_population.observables().timestep.subscribe(rs)
    rs.subscribe(on_next=returns.append)
    steps = []
    scenario.observables().timestep.subscribe(on_next=steps.append)
    # the reference's episode loop, unmodified, to LAST (utils/evaluation/evaluation.py:37-49)
    ns.evaluation.run_episode(focal_population, scenario)
    assert steps[0].step_type.first() and steps[-1].step_type.last()
    assert all(t.step_type.mid() for t in steps[1:-1])
    assert len(steps) == 41                      # FIRST + 40 frames, the last one LAST
    assert len(returns) == 1 and returns[0].shape == (4,)
    assert np.array_equal(returns[0], np.sum([t.reward for t in steps[1:]], axis=0))
    focal_population.close()
  finally:
    scenario.close()


@pytest.mark.gpu
def test_hip_engine_reproduces_the_recorded_scenario_run():
  """The run `test_reference_scenario_on_the_oracle_backed_substrate` recorded — the
  reference's Scenario with scripted background bots on the oracle-backed Substrate — replayed
  on the HIP engine through the same product code (get_factory -> build -> reset / step with
  the joint focal + background actions the Scenario sent): every timestep equal leaf by leaf.
  (The GPU box has no reference tree; what ran there is in the recording.)"""
  with np.load(os.path.join(GOLDEN, "scenario_clean_up_full_timesteps.npz")) as z:
    rec = {k: z[k] for k in z.files}
  env = substrate.get_factory("clean_up").build(SCENARIO["roles"], env_seed=77)
  try:
    full = [env.reset()]
    for a in rec["actions"]:
      full.append(env.step(tuple(int(x) for x in a)))
    got = _record(full, rec["actions"])
    assert set(got) == set(rec)
    for k in sorted(rec):
      assert np.array_equal(got[k], rec[k]), k
    # ... and the partition a Scenario makes of it (scenario.py:63-87) is by player index
    focal = [i for i, f in enumerate(SCENARIO["is_focal"]) if f]
    assert [full[-1].reward[i] for i in focal] == [rec["25.reward"][i] for i in focal]
  finally:
    env.close()
