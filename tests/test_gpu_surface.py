"""GPU tests of the C-ABI surface beyond the step / render parity runs:
debug observations, seeds, streams, player counts, counters, two ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(pack, n, **kw):
  import torch
  from meltingpot_amd import engine
  assert torch.cuda.is_available(), "gpu tests need a GPU"
  return engine.Engine(pack, n, **kw)


def _rollout(eng, oracles, steps, seed, weights=None, each=None):
  import torch
  rng = np.random.default_rng(seed)
  acts = util.random_actions(rng, steps, eng.N, eng.P, eng.num_actions, weights)
  for s in range(steps):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
    if each:
      each(s)


def test_position_and_orientation_observations(clean_up_pack, territory_pack):
  """"N.POSITION" / "N.ORIENTATION" (component_library.lua:806-855: LocationObserver)."""
  from meltingpot_amd import engine as E
  for pack in (clean_up_pack, territory_pack):
    n = 6
    eng = _engine(pack, n)
    oracles = util.make_oracles(pack, n)
    eng.reset()
    for o in oracles:
      o.reset()

    def check(s):
      pos = eng.observe(E.OBS_POSITION).cpu().numpy()
      ori = eng.observe(E.OBS_ORIENTATION).cpu().numpy()
      for w, o in enumerate(oracles):
        _, avat, _ = o.dump()
        assert np.array_equal(pos[w], avat[:, :2]), (s, w)
        assert np.array_equal(ori[w], avat[:, 2]), (s, w)
    check(-1)
    _rollout(eng, oracles, 60, 3, each=check)
    eng.close()


def test_clean_up_debug_metrics_and_zap_matrix(clean_up_pack):
  """clean_up.py:751-784 (_ENABLE_DEBUG_OBSERVATIONS): PLAYER_CLEANED,
  PLAYER_ATE_APPLE, NUM_OTHERS_PLAYER_ZAPPED_THIS_STEP, NUM_OTHERS_WHO_ATE_THIS_STEP;
  playerZapMatrix (avatar_library.lua:657-659).  Engine-owned buffers
  (debug_observations=True) in one engine, caller-bound tensors in another."""
  from meltingpot_amd import engine as E
  pack = util.fertile_clean_up(clean_up_pack, max_rate=0.6)
  n = 8
  kinds = (E.OBS_AUX1, E.OBS_AUX2, E.OBS_AUX3, E.OBS_AUX4)
  for owned in (True, False):
    eng = _engine(pack, n, debug_observations=owned)
    bound = {} if owned else {k: eng.bind(k) for k in kinds + (E.OBS_ZAP_MATRIX,)}
    oracles = util.make_oracles(pack, n)
    eng.reset()
    for o in oracles:
      o.reset()
    seen = np.zeros(5)

    def check(s):
      got = [(bound[k] if not owned else eng.observe(k)).cpu().numpy() for k in kinds]
      zm = (bound[E.OBS_ZAP_MATRIX] if not owned else eng.observe(E.OBS_ZAP_MATRIX)).cpu().numpy()
      for w, o in enumerate(oracles):
        want = o.debug_metrics()
        for k in range(4):
          assert np.array_equal(got[k][w], want[k]), (s, w, k, got[k][w], want[k])
          seen[k] += want[k].sum()
        assert np.array_equal(zm[w], o.zap_matrix()), (s, w)
        seen[4] += o.zap_matrix().sum()
    _rollout(eng, oracles, 250, 17, weights=[1, 6, 2, 2, 2, 1, 1, 5, 5], each=check)
    assert (seen > 0).all(), seen   # every metric was exercised
    eng.close()
  # without either, the debug kinds are not produced
  eng = _engine(pack, 2)
  eng.reset()
  with pytest.raises(E.EngineError, match="not produced"):
    eng.observe(E.OBS_AUX1)
  eng.close()


def test_debug_kinds_other_substrates(commons_pack, territory_pack):
  from meltingpot_amd import engine as E
  eng = _engine(territory_pack, 2)
  with pytest.raises(E.EngineError):
    eng.bind(E.OBS_AUX1)               # territory has no such metrics
  eng.close()
  # commons_harvest: the zap matrix
  n = 4
  eng = _engine(commons_pack, n, debug_observations=True)
  oracles = util.make_oracles(commons_pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  total = [0.0]

  def check(s):
    zm = eng.observe(E.OBS_ZAP_MATRIX).cpu().numpy()
    for w, o in enumerate(oracles):
      assert np.array_equal(zm[w], o.zap_matrix()), (s, w)
      total[0] += o.zap_matrix().sum()
  _rollout(eng, oracles, 150, 5, weights=[1, 4, 1, 1, 1, 2, 2, 6], each=check)
  assert total[0] > 0
  eng.close()


@pytest.mark.parametrize("which", ["clean_up", "commons", "territory"])
def test_layer_observation(clean_up_pack, commons_pack, territory_pack, which):
  """"N.LAYER" int32 [11, 11, L] (avatar_library.lua:246-257; A17)."""
  from meltingpot_amd import engine as E
  pack = {"clean_up": clean_up_pack, "commons": commons_pack, "territory": territory_pack}[which]
  n = 5
  eng = _engine(pack, n)
  oracles = util.make_oracles(pack, n)
  bound = eng.bind(E.OBS_LAYER)   # refreshed by every reset / step (one more small launch)
  eng.reset()
  for o in oracles:
    o.reset()

  def check(s):
    if s % 9:
      return
    lay = eng.observe(E.OBS_LAYER).cpu().numpy()
    assert np.array_equal(lay, bound.cpu().numpy())
    assert lay.shape == (n, eng.P, 11, 11, eng.info.num_layers) and lay.dtype == np.int32
    for w, o in enumerate(oracles):
      for p in range(o.P):
        assert np.array_equal(lay[w, p], o.layer_view(p)), (s, w, p)
  check(0)
  _rollout(eng, oracles, 64, 4, each=check)
  eng.close()


def test_env_seed_and_reset_seeds(clean_up_pack):
  """MpConfig.base_seed != 0 (world w runs seed base + w, builder.py:174-181 with
  one env_seed per world) and mp_reset(seeds=...)."""
  from meltingpot_amd import sharding
  from oracle import oracle
  n, base = 5, 123456789
  eng = _engine(clean_up_pack, n, base_seed=base, world_offset=3)
  oracles = [oracle.Oracle(clean_up_pack, sharding.world_seed(3 + w, base)) for w in range(n)]
  eng.reset()
  for o in oracles:
    o.reset()
  _rollout(eng, oracles, 20, 1)
  grid, avat, glob = eng.dump()
  for w, o in enumerate(oracles):
    og, oa, ogl = o.dump()
    assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa) and np.array_equal(glob[w], ogl)
  # new seeds for worlds 1 and 3 only: their episode count restarts, the others go on
  seeds = np.array([0, 777, 0, 2**63 + 5, 0], np.uint64)
  mask = np.array([0, 1, 0, 1, 0], np.uint8)
  eng.reset(seeds=seeds, mask=mask)
  for w in (1, 3):
    oracles[w] = oracle.Oracle(clean_up_pack, int(seeds[w]))
    oracles[w].reset()
  _rollout(eng, oracles, 20, 2)
  grid, avat, glob = eng.dump()
  for w, o in enumerate(oracles):
    og, oa, ogl = o.dump()
    assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa), w
    assert np.array_equal(glob[w], ogl), (w, glob[w], ogl)
  eng.close()


def test_episodes_of_neighbouring_seeds_do_not_repeat_each_other(clean_up_pack):
  """With the reference's seed + 1 per reset, episode 1 of seed s is episode 0 of
  seed s + 1; a batch seeded base + w would replay its neighbours.  Here (world,
  episode) keys the generator."""
  from meltingpot_amd import engine as E
  pack = util.patch_pack(clean_up_pack, MAXFRAMES=3)
  eng = _engine(pack, 2, base_seed=1000, auto_reset=True)
  eng.reset()
  first = eng.observe(E.OBS_WORLD_RGB).cpu().numpy().copy()    # episode 0 of seeds 1000, 1001
  import torch
  noop = torch.zeros((2, eng.P), dtype=torch.int32, device=eng.device)
  for _ in range(4):   # 3 steps to LAST, the 4th restarts
    eng.step(noop)
  second = eng.observe(E.OBS_WORLD_RGB).cpu().numpy()           # episode 1
  assert eng.dump()[2][0][4] == 2
  assert not np.array_equal(second[0], first[1])   # (1000, ep 1) is not (1001, ep 0)
  assert not np.array_equal(second[0], first[0])
  eng.close()


def test_engine_on_a_non_default_stream(clean_up_pack):
  """mp_set_stream: all work is enqueued on the caller's stream."""
  import torch
  from meltingpot_amd import engine as E
  n = 16
  side = torch.cuda.Stream()
  eng = _engine(clean_up_pack, n)
  oracles = util.make_oracles(clean_up_pack, n)
  with torch.cuda.stream(side):
    eng.use_current_stream()
    rgb = eng.bind(E.OBS_WORLD_RGB)
    eng.reset()
    rng = np.random.default_rng(0)
    acts = util.random_actions(rng, 30, n, eng.P, eng.num_actions)
    dacts = torch.from_numpy(acts).to(eng.device, non_blocking=True)
    for s in range(30):
      eng.step(dacts[s])
    out = rgb.clone()
  side.synchronize()
  for o in oracles:
    o.reset()
  for s in range(30):
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
  out = out.cpu().numpy()
  for w, o in enumerate(oracles):
    assert np.array_equal(out[w], o.render_world()), w
  eng.close()


def test_switching_streams_keeps_the_steps_in_order(clean_up_pack):
  """mp_set_stream orders the new stream after the work already enqueued on the
  old one: a rollout that hops between streams every step — without any host
  synchronisation — equals the oracle's.  `Substrate.step` does that hop itself
  (it follows torch's current stream)."""
  import torch
  from meltingpot_amd import engine as E
  from meltingpot_amd import substrate
  n = 8
  streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()]
  eng = _engine(clean_up_pack, n)
  oracles = util.make_oracles(clean_up_pack, n)
  rng = np.random.default_rng(5)
  acts = util.random_actions(rng, 40, n, eng.P, eng.num_actions)
  dacts = torch.from_numpy(acts).to(eng.device)
  torch.cuda.synchronize()
  eng.reset()
  for s in range(40):
    with torch.cuda.stream(streams[s % 3]):
      eng.use_current_stream()
      eng.step(dacts[s])
  torch.cuda.synchronize()
  for o in oracles:
    o.reset()
    for s in range(40):
      o.step(acts[s, oracles.index(o)])
  grid, avat, glob = eng.dump()
  for w, o in enumerate(oracles):
    og, oa, ogl = o.dump()
    assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa) and np.array_equal(glob[w], ogl), w
  eng.close()
  # the Substrate API follows the caller's stream by itself
  env = substrate.build("clean_up", roles=("default",) * 7, num_worlds=n, env_seed=123)
  ref = substrate.build("clean_up", roles=("default",) * 7, num_worlds=n, env_seed=123)
  env.reset(); ref.reset()
  for s in range(20):
    a = dacts[s, :, :7].contiguous()
    with torch.cuda.stream(streams[s % 2]):
      ts = env.step(a)
    tr = ref.step(a)
  torch.cuda.synchronize()
  assert torch.equal(ts.observation["RGB"], tr.observation["RGB"])
  assert torch.equal(ts.reward, tr.reward)
  env.close(); ref.close()


@pytest.mark.parametrize("name,players", [("clean_up", 3), ("clean_up", 15), ("commons_harvest__open", 7),
                                          ("commons_harvest__open", 1), ("territory__rooms", 4)])
def test_player_count_from_roles(name, players):
  """num_players = len(roles) (configs/substrates/clean_up.py:847): any count up to
  what the pack holds, bit-exact with the oracle run with the same count — state,
  rewards, both views."""
  import torch
  from meltingpot_amd import engine as E
  pack = E.load_pack(name)
  n = 6
  eng = _engine(pack, n, num_players=players)
  assert eng.P == players
  rgb = eng.bind(E.OBS_RGB)
  assert rgb.shape == (n, players, 88, 88, 3)
  oracles = util.make_oracles(pack, n, num_players=players)
  eng.reset()
  for o in oracles:
    o.reset()
  weights = None if name != "commons_harvest__open" else [1, 4, 1, 1, 1, 2, 2, 4]
  _rollout(eng, oracles, 80, players, weights)
  grid, avat, glob = eng.dump()
  rew = eng.observe(E.OBS_REWARD).cpu().numpy()
  wrgb = eng.observe(E.OBS_WORLD_RGB).cpu().numpy()
  got = rgb.cpu().numpy()
  for w, o in enumerate(oracles):
    og, oa, ogl = o.dump()
    assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa) and np.array_equal(glob[w], ogl)
    assert np.array_equal(rew[w], o.rewards())
    assert np.array_equal(wrgb[w], o.render_world())
    for p in range(players):
      assert np.array_equal(got[w, p], o.render_agent(p)), (w, p)
  eng.close()
  with pytest.raises(ValueError):
    E.Engine(pack, 1, num_players=99)


def test_substrate_builds_with_any_number_of_default_roles():
  """The reference's own default (7 roles for commons_harvest__open,
  commons_harvest__open.py:560) and BASELINE.json's 16 both build."""
  from meltingpot_amd import substrate
  for name, counts in (("commons_harvest__open", (7, 16)), ("clean_up", (3, 7))):
    for k in counts:
      with substrate.build(name, roles=("default",) * k, env_seed=5) as env:
        ts = env.reset()
        assert len(ts.reward) == len(ts.observation) == len(env.action_spec()) == k
        ts = env.step([1] * k)
        assert len(ts.reward) == k
  with pytest.raises(ValueError):
    substrate.build("clean_up", roles=("default",) * 16)   # the config has 15 avatar colours
  a = substrate.build("clean_up", roles=("default",) * 7)   # env_seed=None: a random seed
  b = substrate.build("clean_up", roles=("default",) * 7)
  assert a._env_seed != b._env_seed
  a.close(); b.close()


def test_signed_reward_counter(coins_pack):
  """coins pays -2 to the partner of a mismatching collector: the summed reward
  counter is signed."""
  import torch
  from meltingpot_amd import engine as E
  n = 64
  eng = _engine(coins_pack, n)
  eng.reset()
  rng = np.random.default_rng(1)
  total = 0.0
  for s in range(400):
    acts = util.random_actions(rng, 1, n, eng.P, eng.num_actions, [0, 8, 2, 2, 2, 1, 1])[0]
    eng.step(torch.from_numpy(acts).to(eng.device))
    total += float(eng.observe(E.OBS_COLLECTIVE_REWARD).sum())
  c = eng.counters()
  assert c["reward_sum_x1024"] == int(round(total * 1024)), (c, total)
  assert total != 0
  eng.close()


def test_event_rows_hold_a_zap_storm(commons_pack, territory_pack):
  """MP_OBS_EVENTS keeps 127 rows per world-step.  16 commons_harvest players (and
  9 in territory) with half of all actions a beam, 400 steps: nothing is dropped,
  every step's multiset equals the oracle's, and the fullest step stays far below
  the cap."""
  import torch
  from meltingpot_amd import engine as E
  for pack, weights in ((commons_pack, [1, 3, 1, 1, 1, 1, 1, 9]),
                        (territory_pack, [1, 3, 1, 1, 1, 1, 1, 6, 6])):
    n = 16
    eng = _engine(pack, n)
    oracles = util.make_oracles(pack, n)
    eng.reset()
    for o in oracles:
      o.reset()
    rng = np.random.default_rng(3)
    acts = util.random_actions(rng, 400, n, eng.P, eng.num_actions, weights)
    fullest = 0
    for s in range(400):
      eng.step(torch.from_numpy(acts[s]).to(eng.device))
      rows = eng.observe(E.OBS_EVENTS).cpu().numpy()
      assert rows.shape == (n, E.EVENT_ROWS, 4) and E.EVENT_ROWS == 128
      for w, o in enumerate(oracles):
        o.step(acts[s, w])
        cnt = int(rows[w, 0, 0])
        assert rows[w, 0, 1] == 0, "event rows dropped"
        got = sorted(tuple(int(v) for v in r[:3]) for r in rows[w, 1:1 + cnt])
        assert got == o.events(), (s, w)
        fullest = max(fullest, cnt)
    print(f"fullest step: {fullest} events ({eng.P} players)")
    assert 4 <= fullest <= 60, fullest
    eng.close()


@pytest.mark.parametrize("name", ["clean_up", "prisoners_dilemma_in_the_matrix__arena",
                                  "territory__rooms", "coop_mining", "gift_refinements",
                                  "collaborative_cooking__crowded", "externality_mushrooms__dense"])
def test_raw_action_fields(name):
  """mp_step_fields: the raw "<player>.<field>" surface of dmlab2d.Environment.step
  (wrappers/base.py:38-44) — composite actions (move + turn + fire in ONE step,
  level_playing_utils.py:283,333-334) as a device tensor and as a host array,
  fused and stand-alone launch forms, against the oracle fed the same fields;
  ids through mp_step and their ACTION_SET rows through mp_step_fields agree;
  a field outside its actionSpec range is a counted NOOP on the device path and
  a ValueError on the host path."""
  import torch
  from meltingpot_amd import engine as E
  pack = E.load_pack(name)
  n, steps = 12, 120
  eng = _engine(pack, n)
  A = eng.info.num_action_fields
  spec = util.pack_tables(pack)["action_spec"].reshape(-1, 3)
  assert A == len(spec) and A in (3, 4)
  oracles = util.make_oracles(pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(7)
  fields = np.stack([rng.integers(spec[a, 0], spec[a, 1] + 1, size=(steps, n, eng.P))
                     for a in range(A)], axis=-1).astype(np.int32)
  fields[..., 0] = np.where(rng.random((steps, n, eng.P)) < 0.6, fields[..., 0], 0)
  wrgb = None
  composite = 0
  for s in range(steps):
    if s == 40:
      wrgb = eng.bind(E.OBS_WORLD_RGB)     # from here on the fused launch
    if s % 2:
      eng.step_fields(torch.from_numpy(fields[s]).to(eng.device))
    else:
      eng.step_fields(fields[s])
    for w, o in enumerate(oracles):
      o.step_fields(fields[s, w])
    composite += int(np.sum((fields[s, ..., 0] != 0) & (fields[s, ..., 1] != 0) &
                            (fields[s, ..., 2] != 0)))
    if s % 10 == 9:
      grid, avat, glob = eng.dump()
      rew = eng.observe(E.OBS_REWARD).cpu().numpy()
      for w, o in enumerate(oracles):
        og, oa, ogl = o.dump()
        assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa), (s, w)
        assert np.array_equal(glob[w], ogl) and np.array_equal(rew[w], o.rewards()), (s, w)
        if wrgb is not None:
          assert np.array_equal(wrgb[w].cpu().numpy(), o.render_world()), (s, w)
  assert composite > 100 and eng.counters()["bad_actions"] == 0
  # ids == their rows (two fresh engines: the same episode of the same seeds)
  eng.close()
  eng, twin = _engine(pack, n), _engine(pack, n)
  twin.reset(); eng.reset()
  table = util.pack_tables(pack)["action_table"].reshape(-1, 4)[:, :A]
  for s in range(20):
    ids = rng.integers(0, eng.num_actions, size=(n, eng.P), dtype=np.int32)
    eng.step(ids)
    twin.step_fields(np.ascontiguousarray(table[ids]))
  assert all(np.array_equal(a, b) for a, b in zip(eng.dump(), twin.dump()))
  # out of range: device = NOOP + counter (like an id outside ACTION_SET), host = ValueError
  ref = util.make_oracles(pack, n)
  twin.close()
  twin = _engine(pack, n)
  twin.reset()
  for o in ref:
    o.reset()
  bad = np.zeros((n, eng.P, A), np.int32)
  bad[..., 0] = 1                          # everyone forward ...
  bad[0, 0, 0] = 5; bad[1, 1, 1] = 2; bad[2, 0, A - 1] = -1   # ... except three bad ones
  twin.step_fields(torch.from_numpy(bad).to(twin.device))
  assert twin.counters()["bad_actions"] == 3
  grid, avat, _ = twin.dump()
  for w, o in enumerate(ref):
    o.step_fields(bad[w])
    assert np.array_equal(grid[w], o.dump()[0]) and np.array_equal(avat[w], o.dump()[1]), w
  with pytest.raises(ValueError):
    twin.step_fields(bad)
  twin.close()
  eng.close()


def test_substrate_with_a_custom_action_table():
  """Substrate(..., action_table=...) — build_substrate's parameter
  (utils/substrates/substrate.py:107-139) — batched on the device and one world
  on the host; a table outside the action spec is refused where the reference
  refuses it (discrete_action_wrapper.py:28-49)."""
  import torch
  from meltingpot_amd import engine as E, substrate
  table = ({"move": 0, "turn": 0, "fireZap": 0},
           {"move": 1, "turn": 1, "fireZap": 0},
           {"move": 2, "turn": -1, "fireZap": 1},
           {"move": 3, "turn": 1, "fireZap": 1})
  rows = np.array([[r["move"], r["turn"], r["fireZap"]] for r in table], np.int32)
  roles = ("default",) * 5
  with substrate.build("commons_harvest__open", roles=roles, num_worlds=6, env_seed=77,
                       action_table=table) as env:
    assert env.action_spec()[0].num_values == 4
    oracles = [__import__("oracle.oracle", fromlist=["x"]).Oracle(
        E.load_pack("commons_harvest__open"), 77 + w, 5) for w in range(6)]
    for o in oracles:
      o.reset()
    env.reset()
    rng = np.random.default_rng(2)
    for s in range(40):
      a = rng.integers(0, 4, size=(6, 5))
      ts = env.step(torch.from_numpy(a).to(env.engine.device) if s % 2 else a)
      for w, o in enumerate(oracles):
        o.step_fields(rows[a[w]])
    rgb = ts.observation["RGB"].cpu().numpy()
    for w, o in enumerate(oracles):
      for p in range(5):
        assert np.array_equal(rgb[w, p], o.render_agent(p)), (w, p)
    with pytest.raises(ValueError):
      env.step(np.full((6, 5), 4))
  with pytest.raises(ValueError):
    substrate.build("commons_harvest__open", roles=roles, action_table=[{"move": 7, "turn": 0,
                                                                        "fireZap": 0}])
  with pytest.raises(ValueError):
    substrate.build("commons_harvest__open", roles=roles, action_table=[{"move": 1}])
  with substrate.build("commons_harvest__open", roles=roles, env_seed=5, action_table=table) as one:
    one.reset()
    ts = one.step([3, 2, 1, 0, 3])
    assert len(ts.observation) == 5 and ts.observation[0]["RGB"].shape == (88, 88, 3)


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_two_ranks_two_engines_through_bench(tmp_path, launcher):
  """bench.py's rank path: two processes, one engine each (both on this one GPU,
  gloo for the window reduction), worlds sharded by global index — launched the
  driver's way (torch.distributed.run around bench.py) and as plain
  `python bench.py --gpus 2` (bench.py starts its own ranks)."""
  env = dict(os.environ, MASTER_ADDR="127.0.0.1")
  for k in list(env):
    if k.startswith("MP_RENDER_") or k == "MP_ENGINE_LIB" or k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
      del env[k]
  cmd = [sys.executable]
  if launcher == "torchrun":
    cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
            "--master-addr", "127.0.0.1", "--master-port", "29531"]
  cmd += [os.path.join(ROOT, "bench.py"),
          "--gpus", "2", "--steps", "8", "--warmup", "2", "--worlds", "96", "--no-cpu-baseline",
          "--no-traffic", "--one-device", "--dist-backend", "gloo"]
  out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
  line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
  assert line["n_gpus"] == 2 and line["scaling"] == "weak"
  assert line["counters"]["world_steps"] == 2 * 96 * 10      # both ranks, warm-up included
  assert line["counters"]["agent_steps"] == 2 * 96 * 10 * 7
  assert line["value"] > 0
  ranks = line["ranks"]
  assert ranks["count"] == 2 and len(ranks["devices"]) == 2
  assert all(d.startswith("cuda:0") for d in ranks["devices"])
  assert all(ms > 0 for ms in ranks["ms_per_step"])
  assert max(ranks["ms_per_step"]) <= line["ms_per_step"] * 1.0001


def test_eight_engine_bearing_ranks_on_one_gpu():
  """What the driver's 8-GPU run does, as far as one GPU can show it: EIGHT ranks started the
  driver's way (torch.distributed.run around bench.py), each with its own engine, its own
  shard of the 8 x 256 worlds, its own placement probe (--place 1: eight probes at once on
  one device would time each other) — gloo for the window, every rank on device 0.  The
  line must count eight ranks and the counters must add up over the eight shards."""
  env = dict(os.environ, MASTER_ADDR="127.0.0.1")
  for k in list(env):
    if k == "MP_ENGINE_LIB" or k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
      del env[k]
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
         "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
         "--gpus", "8", "--one-device", "--dist-backend", "gloo", "--worlds", "256",
         "--steps", "5", "--warmup", "2", "--place", "1", "--no-cpu-baseline", "--no-traffic"]
  out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
  line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
  assert line["n_gpus"] == 8 and line["scaling"] == "weak"
  ranks = line["ranks"]
  assert ranks["count"] == 8 and len(ranks["devices"]) == 8 and ranks["backend"] == "gloo"
  assert all(ms > 0 for ms in ranks["ms_per_step"])
  assert line["counters"]["world_steps"] == 8 * 256 * 7        # eight shards, warm-up included
  assert line["counters"]["agent_steps"] == 8 * 256 * 7 * 7
  assert line["value"] > 0 and line["config"]["worlds_per_gpu"] == 256


def test_window_reduction_over_rccl(tmp_path):
  """The two all-reduces of a measurement window and the rank evidence carried
  by RCCL itself (backend "nccl"; one rank: a 1-GPU box cannot hold two), fed
  with a live engine's counters."""
  script = tmp_path / "rccl_window.py"
  script.write_text(f"""
import os, sys, json
sys.path.insert(0, {ROOT!r})
import torch, torch.distributed as dist
import bench
from meltingpot_amd import engine as E, sharding
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
eng = E.Engine(E.load_pack("clean_up"), 32, device=0)
eng.reset()
acts = torch.zeros((32, eng.P), dtype=torch.int32, device=eng.device)
for _ in range(5):
  eng.step(acts)
class _Two:   # reduce_window skips the collective for one rank: present two
  def __getattr__(self, k): return getattr(dist, k)
  def get_world_size(self): return 2
secs, tot = sharding.reduce_window(1.5, eng.counters(), E.COUNTER_NAMES, _Two(), eng.device)
ev = bench._rank_evidence(dist, "cuda:0", eng.device, 0.25)
print(json.dumps({{"secs": secs, "tot": tot, "ev": ev}}))
eng.close()
dist.destroy_process_group()
""")
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
  for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
    env.pop(k, None)
  out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
  assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
  got = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
  assert got["secs"] == 1.5 and got["tot"]["world_steps"] == 32 * 5
  assert got["ev"] == {"count": 1, "backend": "nccl", "devices": ["cuda:0"], "ms_per_step": [0.25]}


def test_a_bound_view_is_placed_without_touching_the_state(clean_up_pack):
  """Engine.bind() without a tensor allocates a large pixel view through
  mp_place_output: several candidate buffers (mapped from 2 MB physical chunks), the
  engine's own launch timed DRY on each (a reset that names no world) under the plan
  that suits it (mp_tune), the fastest kept (profiles/r04_write_fronts.md).  The probe
  must leave state, scalar outputs and episode counters alone, and the rollout that
  follows is the oracle's."""
  import torch
  from meltingpot_amd import engine as E
  n = 640                                          # WORLD.RGB: 77 MB, above PLACE_MIN_BYTES
  eng = E.Engine(clean_up_pack, n, placements=5)
  eng.reset()
  rng = np.random.default_rng(3)
  acts = util.random_actions(rng, 12, n, eng.P, eng.num_actions)
  for s in range(6):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
  before = eng.dump()
  rew = eng.observe(E.OBS_REWARD).cpu().numpy().copy()
  counters = eng.counters()
  wrgb = eng.bind(E.OBS_WORLD_RGB)                 # five candidates, probed dry
  info = eng.placement[E.OBS_WORLD_RGB]
  assert info["probe"] == "dry"     # (an engine in use)
  assert 2 <= info["candidates"] <= 5 and len(info["dry_launch_us"]) == info["candidates"]
  # (every fourth candidate is one plain allocation, the others are mapped from 2 MB chunks)
  assert info["kind"] == ("plain allocation" if info["picked"] == 3 else "mapped 2 MB")
  assert info["dry_launch_us"][info["picked"]] == min(info["dry_launch_us"])   # (rounded: ties)
  # the library's memory: torch's allocator was not asked for the view
  assert wrgb.data_ptr() not in {b.data_ptr() for b in [eng.empty(E.OBS_REWARD)]}
  after = eng.dump()
  for a, b in zip(before, after):
    assert np.array_equal(a, b)
  assert np.array_equal(rew, eng.observe(E.OBS_REWARD).cpu().numpy())
  assert counters == eng.counters()
  # a small view, or a caller's own tensor, is not probed
  eng.bind(E.OBS_REWARD)
  assert E.OBS_REWARD not in eng.placement
  # ... and the engine goes on as the oracle does
  sample = [0, 1, 317, n - 1]
  oracles = [util.make_oracles(clean_up_pack, 1, offset=w)[0] for w in sample]
  for o in oracles:
    o.reset()
    for s in range(6):
      o.step(acts[s, sample[oracles.index(o)]])
  for s in range(6, 12):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in zip(sample, oracles):
      o.step(acts[s, w])
  got = wrgb.cpu().numpy()
  grid = eng.dump()[0]
  for w, o in zip(sample, oracles):
    assert np.array_equal(grid[w], o.dump()[0]), w
    assert np.array_equal(got[w], o.render_world()), w
  eng.close()


def test_two_placed_views_are_the_memory_the_launch_writes(clean_up_pack):
  """Both pixel views placed one after the other (what `substrate.build(...,
  num_worlds=N)` does): the second placement's candidates used to reuse the virtual
  ranges the first one's losers had freed, and the launch then wrote through stale
  translations — 267 - 1030 of 1030 worlds of the second view came back stale.  The
  library now retires a mapped view's range with it (mp_free_output)."""
  import torch
  from meltingpot_amd import engine as E
  n = 1030
  eng = E.Engine(clean_up_pack, n, placements=6)
  rgb = eng.bind(E.OBS_RGB)
  wrgb = eng.bind(E.OBS_WORLD_RGB)
  assert set(eng.placement) == {E.OBS_RGB, E.OBS_WORLD_RGB} and eng.fused
  oracles = util.make_oracles(clean_up_pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(12)
  acts = util.random_actions(rng, 3, n, eng.P, eng.num_actions)
  for s in range(3):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
  a, b = rgb.cpu().numpy(), wrgb.cpu().numpy()
  for w, o in enumerate(oracles):
    assert np.array_equal(b[w], o.render_world()), w
    for p in range(eng.P):
      assert np.array_equal(a[w, p], o.render_agent(p)), (w, p)
  # a third placement (rebinding) after the first two were dropped
  eng.unbind(E.OBS_RGB)
  del rgb, a
  rgb = eng.bind(E.OBS_RGB)
  eng.step(torch.from_numpy(acts[0]).to(eng.device))
  for w, o in enumerate(oracles):
    o.step(acts[0, w])
  a = rgb.cpu().numpy()
  for w in (0, 1, 515, n - 1):
    for p in range(eng.P):
      assert np.array_equal(a[w, p], oracles[w].render_agent(p)), (w, p)
  eng.close()


def test_a_view_mapped_from_physical_chunks(commons_pack):
  """mp_alloc_output(chunk_bytes > 0): one virtual range mapped onto separate
  physical chunks — what mp_place_output's candidates are made of.
  Such a tensor is a bound view like any other (the fused launch writes it, bit-exact
  vs the oracle) and its memory goes back with it."""
  import gc
  import torch
  from meltingpot_amd import engine as E
  n = 40
  eng = E.Engine(commons_pack, n)
  before = torch.cuda.mem_get_info()[0]
  rgb = eng.empty_mapped(E.OBS_RGB, 2 << 20)
  assert rgb is not None and tuple(rgb.shape) == eng.shapes[E.OBS_RGB][0]
  assert torch.cuda.mem_get_info()[0] < before
  eng.bind(E.OBS_RGB, rgb)
  oracles = util.make_oracles(commons_pack, n)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(8)
  acts = util.random_actions(rng, 25, n, eng.P, eng.num_actions)
  for s in range(25):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
  got = rgb.cpu().numpy()
  for w, o in enumerate(oracles):
    for p in range(eng.P):
      assert np.array_equal(got[w, p], o.render_agent(p)), (w, p)
  eng.unbind(E.OBS_RGB)
  del rgb, got
  gc.collect()
  torch.cuda.synchronize()
  assert torch.cuda.mem_get_info()[0] >= before - (4 << 20)
  eng.close()


def test_placing_a_view_on_an_untouched_engine_leaves_no_trace(clean_up_pack):
  """An engine nothing has been done with is probed by REAL steps behind a device-side
  copy of its records and counters (mp_tune): afterwards it must be the engine it was — the first reset
  starts episode 0 with the oracle's draws, the counters start from nothing.  A
  caller's own tensor is not placed, its plan is tuned (mp_tune), to the same end."""
  import torch
  from meltingpot_amd import engine as E
  n = 640
  eng = E.Engine(clean_up_pack, n, placements=3)
  wrgb = eng.bind(E.OBS_WORLD_RGB)
  info = eng.placement[E.OBS_WORLD_RGB]
  assert info["probe"] == "stepped behind a copy" and 2 <= info["candidates"] <= 3
  own = torch.empty_like(wrgb)
  eng.bind(E.OBS_WORLD_RGB, own)       # (the placed one goes back to the driver)
  assert eng.tune() > 0.0
  wrgb = own
  assert eng.counters()["world_steps"] == 0 and eng.counters()["episodes"] == 0
  sample = [0, 5, 333, n - 1]
  oracles = [util.make_oracles(clean_up_pack, 1, offset=w)[0] for w in sample]
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(4)
  acts = util.random_actions(rng, 10, n, eng.P, eng.num_actions)
  for s in range(10):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in zip(sample, oracles):
      o.step(acts[s, w])
  grid, avat, glob = eng.dump()
  got = wrgb.cpu().numpy()
  for w, o in zip(sample, oracles):
    og, oa, ogl = o.dump()
    assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa) and np.array_equal(glob[w], ogl), w
    assert np.array_equal(got[w], o.render_world()), w
  assert eng.counters()["world_steps"] == 10 * n and eng.counters()["episodes"] == n
  eng.close()


@pytest.mark.gpu
def test_box_fill_times_the_bound_view_and_touches_nothing_else(clean_up_pack):
  """mp_box_fill (the bench line's calibration): three fill rates of the buffer bound for a
  pixel view — every byte of the view is overwritten by each of the store loops (a sentinel
  survives nowhere), the engine's records and the other view are what they were, the next
  step redraws the view bit-exact; kinds that are not one bound pixel buffer are refused."""
  import torch
  from meltingpot_amd import engine as E
  n = 300     # ragged against the plan's workgroups: the last workgroup's range is short
  eng = _engine(clean_up_pack, n)
  wrgb = eng.bind(E.OBS_WORLD_RGB)
  rgb = eng.bind(E.OBS_RGB)
  oracles = util.make_oracles(clean_up_pack, 3)
  eng.reset()
  for o in oracles:
    o.reset()
  rng = np.random.default_rng(5)
  acts = util.random_actions(rng, 4, n, eng.P, eng.num_actions)
  eng.step(torch.from_numpy(acts[0]).to(eng.device))
  before = eng.dump()
  keep_rgb = rgb.clone()
  with pytest.raises(ValueError, match="not a pixel view"):
    eng.box_fill(E.OBS_REWARD)
  wrgb.fill_(7)
  rep = eng.box_fill(E.OBS_WORLD_RGB, reps=3)
  assert rep["bytes"] == wrgb.numel() and rep["span_bytes"] == 2 * 8 * 30 * 24
  assert all(rep[k] > 0 for k in ("memset_us", "product_order_us", "front_4k_us"))
  assert rep["workgroups"] >= 1 and rep["storing_waves"] >= 1
  # the last loop was the 4 KiB front: its junk covers the whole view (no byte keeps the 7s
  # pattern over 16 bytes: every 16-byte chunk holds the loop's {ticket, offset, 2, 3})
  chunks = wrgb.view(-1).view(torch.int32).view(-1, 4)
  assert bool(((chunks[:, 2] == 2) & (chunks[:, 3] == 3)).all())
  assert torch.equal(rgb, keep_rgb)
  after = eng.dump()
  for a, b in zip(before, after):
    assert np.array_equal(a, b)
  eng.unbind(E.OBS_WORLD_RGB)
  with pytest.raises(ValueError, match="not bound"):
    eng.box_fill(E.OBS_WORLD_RGB)
  eng.bind(E.OBS_WORLD_RGB, wrgb)
  for s in range(1, 4):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
  for w, o in enumerate(oracles):
    for s in range(4):
      o.step(acts[s, w])
    assert np.array_equal(wrgb[w].cpu().numpy(), o.render_world())
    assert np.array_equal(rgb[w].cpu().numpy(), np.stack([o.render_agent(p) for p in range(o.P)]))
  eng.close()
