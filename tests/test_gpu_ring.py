"""GPU: the rollout ring (mp_bind_output_ring, `Substrate(rollout_length=T)`) and the error
paths of mp_tune / mp_place_output / mp_bind_output, against the CPU oracle.

Reference behaviour served by the ring: the reference hands back FRESH arrays every step
(utils/substrates/wrappers/multiplayer_wrapper.py:108-118, utils/substrates/substrate.py:74-81),
a rollout just stores them.  Here slot t % T of a [T, N, ...] tensor holds step t."""
import ctypes
import time

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _engine(pack, n, **kw):
  import torch
  from meltingpot_amd import engine
  assert torch.cuda.is_available(), "gpu tests need a GPU"
  return engine.Engine(pack, n, **kw)


def _oracle_obs(o):
  return {"world": o.render_world(), "agents": np.stack([o.render_agent(p) for p in range(o.P)]),
          "reward": o.rewards().copy(), "ready": o.ready_to_shoot().copy()}


@pytest.mark.parametrize("which,n,T,dev", [
    ("clean_up", 40, 5, None),
    ("clean_up", 70, 3, {"batch_worlds": 1, "ring_batches": 6, "static_pct": 50, "max_groups": 4}),
    ("commons", 30, 4, None),
    ("territory", 20, 2, None),
    ("clean_up", 9, 1, None),          # a ring of one slot is a bound buffer
])
def test_slot_t_holds_step_t(clean_up_pack, commons_pack, territory_pack, which, n, T, dev):
  """Submission t (the reset is submission 0) writes slot t % T of every ring-bound kind —
  both pixel views from the ONE fused launch, rewards, READY_TO_SHOOT, step type — and
  leaves the other slots alone: after every step each slot is compared with what the
  oracle said at the step it belongs to."""
  import torch
  from meltingpot_amd import engine as E
  pack = {"clean_up": clean_up_pack, "commons": commons_pack, "territory": territory_pack}[which]
  eng = _engine(pack, n, dev=dev)
  rings = {k: eng.bind_ring(k, slots=T) for k in
           (E.OBS_RGB, E.OBS_WORLD_RGB, E.OBS_REWARD, E.OBS_READY_TO_SHOOT, E.OBS_STEP_TYPE)}
  assert eng.fused and eng.ring == {"slots": T, "next": 0, "last": T - 1}
  oracles = util.make_oracles(pack, n)
  history = []          # per submission: per world oracle observations

  def record(first):
    history.append([dict(_oracle_obs(o), reward=np.zeros(o.P) if first else o.rewards().copy())
                    for o in oracles])

  def check(tag):
    t_last = len(history) - 1
    assert eng.ring["last"] == t_last % T and eng.ring["next"] == (t_last + 1) % T
    for t in range(max(0, t_last - T + 1), t_last + 1):     # every submission a slot still holds
      s = t % T
      got_w = rings[E.OBS_WORLD_RGB][s].cpu().numpy()
      got_a = rings[E.OBS_RGB][s].cpu().numpy()
      got_r = rings[E.OBS_REWARD][s].cpu().numpy()
      got_y = rings[E.OBS_READY_TO_SHOOT][s].cpu().numpy()
      for w in range(n):
        want = history[t][w]
        assert np.array_equal(got_w[w], want["world"]), (tag, t, s, w, "WORLD.RGB")
        assert np.array_equal(got_a[w], want["agents"]), (tag, t, s, w, "RGB")
        assert np.array_equal(got_r[w], want["reward"]), (tag, t, s, w, "REWARD")
        assert np.array_equal(got_y[w], want["ready"]), (tag, t, s, w, "READY_TO_SHOOT")
      assert (rings[E.OBS_STEP_TYPE][s].cpu().numpy() == (0 if t == 0 else 1)).all()
    # mp_observe of a ring-bound scalar kind reads the slot written last
    assert np.array_equal(eng.observe(E.OBS_REWARD).cpu().numpy(),
                          np.stack([h["reward"] for h in history[t_last]]))

  eng.reset()
  for o in oracles:
    o.reset()
  record(True)
  check("reset")
  rng = np.random.default_rng(T * 100 + n)
  steps = 2 * T + 3
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions)
  for s in range(steps):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
    record(False)
    check(f"step {s + 1}")
  assert not eng.fault_words()[:6].any()
  # a kind leaves the ring with mp_bind_output; the others keep their position
  plain = eng.bind(E.OBS_REWARD)
  eng.step(torch.from_numpy(acts[0]).to(eng.device))
  for w, o in enumerate(oracles):
    o.step(acts[0, w])
  assert np.array_equal(plain.cpu().numpy(), np.stack([o.rewards() for o in oracles]))
  assert eng.ring["slots"] == T
  with pytest.raises(ValueError, match="slots"):
    eng.bind_ring(E.OBS_AUX0 if which == "clean_up" else E.OBS_DISCOUNT, slots=T + 1)
  eng.close()


def test_ring_argument_errors(clean_up_pack):
  import torch
  from meltingpot_amd import engine as E
  eng = _engine(clean_up_pack, 8)
  L, h = eng._L, eng._h
  buf = torch.empty(4 * 8 * 7 * 8 + 4096, dtype=torch.uint8, device=eng.device)
  bind = lambda kind, ptr, stride, slots: L.mp_bind_output_ring(h, kind, ptr, stride, slots)
  assert bind(E.OBS_REWARD, buf.data_ptr(), 8 * 7 * 8, 4) == E.MP_ERR_INVALID     # 448: not a multiple of 256
  assert bind(E.OBS_REWARD, buf.data_ptr(), 256, 4) == E.MP_ERR_INVALID           # does not hold 448 bytes
  assert bind(E.OBS_REWARD, buf.data_ptr(), 512, 0) == E.MP_ERR_INVALID
  assert bind(E.OBS_REWARD, buf.data_ptr(), 512, 4) == 0
  assert bind(E.OBS_AUX0, buf.data_ptr(), 512, 5) == E.MP_ERR_INVALID             # one slot count for all
  assert bind(E.OBS_INVENTORY, buf.data_ptr(), 512, 4) == -5                      # clean_up has none
  assert bind(E.OBS_REWARD, buf.data_ptr(), 1 << 20, 1 << 20) == E.MP_ERR_INVALID   # 1 TiB: runs off the allocation
  host = np.zeros(4096, np.uint8)
  assert bind(E.OBS_REWARD, host.ctypes.data, 512, 4) == E.MP_ERR_INVALID         # host memory
  assert b"host memory" in L.mp_last_error() or b"not memory the device" in L.mp_last_error()
  assert L.mp_bind_output(h, E.OBS_REWARD, host.ctypes.data) == E.MP_ERR_INVALID
  assert bind(E.OBS_REWARD, None, 0, 0) == 0 and eng.ring["slots"] == 0           # NULL unbinds
  eng.close()


def test_substrate_rollout_length(clean_up_pack):
  """`substrate.build(..., num_worlds=N, rollout_length=T)`: every TimeStep's leaves are views
  of slot `.slot` of `env.rollout`'s [T, N, ...] tensors, they stay what they were for T
  steps, and they are the oracle's observations."""
  import torch
  from meltingpot_amd import substrate
  n, T = 24, 6
  cfg = substrate.get_config("clean_up")
  env = substrate.build("clean_up", roles=cfg.default_player_roles, num_worlds=n,
                        rollout_length=T, env_seed=900)
  from oracle import oracle as oracle_lib
  oracles = [oracle_lib.Oracle(clean_up_pack, 900 + w, 7) for w in range(n)]
  ro = env.rollout
  assert ro["observation"]["RGB"].shape == (T, n, 7, 88, 88, 3)
  assert ro["observation"]["WORLD.RGB"].shape == (T, n, 168, 240, 3)
  assert ro["reward"].shape == (T, n, 7) and ro["step_type"].shape == (T, n)
  kept = []
  ts = env.reset()
  for o in oracles:
    o.reset()
  assert isinstance(ts, substrate.TimeStep) and ts.slot == 0 == env.slot
  kept.append((ts, [_oracle_obs(o) for o in oracles]))
  rng = np.random.default_rng(2)
  for s in range(T - 1):
    a = rng.integers(0, 9, size=(n, 7)).astype(np.int32)
    ts = env.step(torch.from_numpy(a).to(env.engine.device))
    for w, o in enumerate(oracles):
      o.step(a[w])
    assert ts.slot == s + 1 == env.engine.ring["last"]
    assert ts.observation["RGB"].data_ptr() == ro["observation"]["RGB"][ts.slot].data_ptr()
    step_type, reward, discount, observation = ts            # it IS a TimeStep
    assert (step_type.cpu().numpy() == 1).all() and (discount.cpu().numpy() == 1).all()
    kept.append((ts, [_oracle_obs(o) for o in oracles]))
  # nothing the learner kept was overwritten: all T timesteps are still their own step
  for ts, want in kept:
    rgb = ts.observation["RGB"].cpu().numpy()
    wrgb = ts.observation["WORLD.RGB"].cpu().numpy()
    rdy = ts.observation["READY_TO_SHOOT"].cpu().numpy()
    for w in range(n):
      assert np.array_equal(rgb[w], want[w]["agents"]) and np.array_equal(wrgb[w], want[w]["world"])
      assert np.array_equal(rdy[w], want[w]["ready"])
    if ts.slot:
      assert np.array_equal(ts.reward.cpu().numpy(), np.stack([x["reward"] for x in want]))
  # ... and the next step reuses slot 0
  a = rng.integers(0, 9, size=(n, 7)).astype(np.int32)
  ts = env.step(torch.from_numpy(a).to(env.engine.device))
  assert ts.slot == 0
  # _replace keeps the slot (it IS a NamedTuple: a fresh tuple would have lost it)
  assert ts._replace(reward=None).slot == 0 and ts._replace(reward=None).reward is None
  # a submission made through the ENGINE (a masked reset of one world) moves the ring on, and
  # the substrate's slot follows the engine's position, not a count of its own calls
  mask = np.zeros(n, np.uint8); mask[3] = 1
  env.engine.reset(mask=mask)
  assert env.slot == 1 == env.engine.ring["last"]
  ts = env.step(torch.from_numpy(a).to(env.engine.device))
  assert ts.slot == 2 and ts.observation["RGB"].data_ptr() == ro["observation"]["RGB"][2].data_ptr()
  with pytest.raises(ValueError, match="batched"):
    substrate.build("clean_up", roles=cfg.default_player_roles, rollout_length=4)
  env.close()


def test_ring_at_full_size_keeps_a_plan_per_slot_and_never_synchronises(clean_up_pack):
  """4096 clean_up worlds, per-agent RGB in a ring of 4 slots (2.7 GB): mp_tune times every
  slot once at bind time; afterwards a step is a pointer store + one launch — the host gets
  through 32 steps in a fraction of the time the GPU needs for them (nothing synchronises,
  nothing re-tunes) — and sampled worlds of every slot are the oracle's."""
  import torch
  from meltingpot_amd import engine as E
  n, T, steps = 4096, 4, 32
  eng = _engine(clean_up_pack, n)
  ring = eng.bind_ring(E.OBS_RGB, slots=T)        # (tunes: every slot timed, a plan kept per slot)
  assert ring.shape[0] == T and eng.fused
  plan = eng.plan
  assert plan["batch_worlds"] >= 1 and plan["workgroups"] >= 1
  eng.reset()
  gen = torch.Generator(device=eng.device)
  gen.manual_seed(1)
  acts = torch.randint(0, eng.num_actions, (steps, n, eng.P), generator=gen, device=eng.device,
                       dtype=torch.int32)
  eng.sync()
  t0 = time.perf_counter()
  for s in range(steps):
    eng.step(acts[s])
  host_s = time.perf_counter() - t0
  eng.sync()
  total_s = time.perf_counter() - t0
  # one launch is > 150 us of GPU time, enqueueing it ~10 - 30 us of host time
  assert host_s < 0.5 * total_s, (host_s, total_s)
  assert eng.ring["last"] == steps % T          # reset + 32 steps = 33 submissions -> slot 0
  sample = [0, 1, 2047, 2048, n - 1, 777]
  host_acts = acts.cpu().numpy()
  for w in sample:
    o = util.make_oracles(clean_up_pack, 1, offset=w)[0]
    o.reset()
    for s in range(steps):
      o.step(host_acts[s, w])
      t = s + 1
      if t > steps - T:     # the last T submissions are what the ring holds
        want = np.stack([o.render_agent(p) for p in range(o.P)])
        assert np.array_equal(ring[t % T, w].cpu().numpy(), want), (w, t)
  assert not eng.fault_words()[:6].any()
  eng.close()


# --------------------------------------------------------------------------
# error paths of mp_tune / mp_place_output


def _state(eng):
  from meltingpot_amd import engine as E
  grid, avat, glob = eng.dump()
  return (grid.tobytes(), avat.tobytes(), glob.tobytes(), tuple(sorted(eng.counters().items())),
          eng.observe(E.OBS_STEP_TYPE).cpu().numpy().tobytes(),
          eng.observe(E.OBS_REWARD).cpu().numpy().tobytes(),
          eng.observe(E.OBS_EVENTS).cpu().numpy().tobytes())


def test_probe_leaves_a_fresh_engine_and_the_callers_buffers_alone(clean_up_pack):
  """mp_tune / mp_place_output on an engine nothing has been done with REALLY step it (behind
  a copy): afterwards records, counters, the engine's own scalar outputs and the caller's
  bound scalar buffers are bit-identical to before — then the engine plays as if never probed."""
  import torch
  from meltingpot_amd import engine as E
  n = 600
  eng = _engine(clean_up_pack, n)
  reward = torch.full((n, 7), 123.0, dtype=torch.float64, device=eng.device)
  ready = torch.full((n, 7), -5.0, dtype=torch.float64, device=eng.device)
  eng.bind(E.OBS_REWARD, reward)
  eng.bind(E.OBS_READY_TO_SHOOT, ready)
  before = _state(eng)
  view = eng.place(E.OBS_WORLD_RGB, candidates=3)      # 72 MB x 3, stepped probe
  assert eng.placement[E.OBS_WORLD_RGB]["probe"] == "stepped behind a copy"
  assert eng.placement[E.OBS_WORLD_RGB]["setup_s"] > 0
  assert eng.placement[E.OBS_WORLD_RGB]["requested"] == 3
  us = eng.tune()
  assert us > 0
  assert _state(eng) == before
  assert (reward == 123.0).all() and (ready == -5.0).all()
  oracles = util.make_oracles(clean_up_pack, 4)
  eng.reset()
  rng = np.random.default_rng(0)
  acts = util.random_actions(rng, 10, n, eng.P, eng.num_actions)
  for o in oracles:
    o.reset()
  for s in range(10):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
    for w, o in enumerate(oracles):
      o.step(acts[s, w])
  for w, o in enumerate(oracles):
    assert np.array_equal(view[w].cpu().numpy(), o.render_world())
    assert np.array_equal(reward[w].cpu().numpy(), o.rewards())
  eng.close()


def test_failed_placement_leaves_the_engine_bit_identical(clean_up_pack):
  """Injected failures: a max_bytes that holds no view (MP_ERR_INVALID), a bound on retired
  address space that refuses every mapping (MP_ERR_HIP, with the reason), a kind that is no
  pixel view — after each the engine is what it was, the kind is bound to what it was bound
  to, and `Engine.bind` falls back to an ordinary allocation instead of failing."""
  import torch
  from meltingpot_amd import engine as E
  n = 900     # WORLD.RGB = 109 MB: above PLACE_MIN_BYTES
  eng = _engine(clean_up_pack, n)
  mine = eng.empty(E.OBS_WORLD_RGB)
  eng.placements = 0
  eng.bind(E.OBS_WORLD_RGB, mine)
  eng.placements = 24
  eng.reset()
  rng = np.random.default_rng(3)
  acts = util.random_actions(rng, 6, n, eng.P, eng.num_actions)
  for s in range(3):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
  before, pixels = _state(eng), mine.clone()
  with pytest.raises(ValueError, match="holds no view"):
    eng.place(E.OBS_WORLD_RGB, max_bytes=1 << 20)
  with pytest.raises(ValueError, match="not a pixel view"):
    eng.place(E.OBS_REWARD)
  limit = eng.retired_va
  assert limit["limit"] == 16 << 40
  try:
    assert eng._L.mp_set_retired_va_limit(limit["bytes"]) == 0      # nothing more may be retired
    with pytest.raises(E.EngineError, match="retired"):
      eng.place(E.OBS_WORLD_RGB)
    assert _state(eng) == before
    # the engine is in use: a dry probe, which only DRAWS the bound view (the same pixels)
    assert torch.equal(mine, pixels)
    eng.step(torch.from_numpy(acts[3]).to(eng.device))             # still bound to `mine`
    assert not torch.equal(mine, pixels)
    # bind() without a tensor: placing fails, an ordinary allocation takes its place
    eng.unbind(E.OBS_WORLD_RGB)
    fallback = eng.bind(E.OBS_WORLD_RGB)
    assert "placing failed" in eng.placement[E.OBS_WORLD_RGB]["kind"]
    eng.step(torch.from_numpy(acts[4]).to(eng.device))
  finally:
    assert eng._L.mp_set_retired_va_limit(limit["limit"]) == 0
  oracles = util.make_oracles(clean_up_pack, 3)
  for o in oracles:
    o.reset()
    for s in range(5):
      o.step(acts[s, oracles.index(o)])
  for w, o in enumerate(oracles):
    assert np.array_equal(fallback[w].cpu().numpy(), o.render_world())
  # retiring: a placed view given back grows the process's retired range by its size
  r0 = eng.retired_va["bytes"]
  eng.unbind(E.OBS_WORLD_RGB)
  placed = eng.place(E.OBS_WORLD_RGB, candidates=2)
  assert eng.retired_va["bytes"] >= r0 + placed.numel()      # the loser was released
  eng.close()


def test_tune_between_two_steps(clean_up_pack, commons_pack):
  """mp_tune while a step is in flight: it lets the step finish, probes dry (an engine in
  use is never stepped by a probe) and the run goes on bit-exact."""
  import torch
  from meltingpot_amd import engine as E
  for pack, kind, n in ((clean_up_pack, E.OBS_WORLD_RGB, 700), (commons_pack, E.OBS_RGB, 300)):
    eng = _engine(pack, n, placements=0)
    view = eng.bind(kind)
    eng.reset()
    rng = np.random.default_rng(n)
    acts = util.random_actions(rng, 12, n, eng.P, eng.num_actions)
    dacts = torch.from_numpy(acts).to(eng.device)
    for s in range(12):
      eng.step(dacts[s])
      if s in (2, 3, 8):
        eng.tune()          # no sync in between: the step above is still running
    oracles = util.make_oracles(pack, 5)
    for w, o in enumerate(oracles):
      o.reset()
      for s in range(12):
        o.step(acts[s, w])
    grid, avat, glob = eng.dump()
    for w, o in enumerate(oracles):
      og, oa, ogl = o.dump()
      assert np.array_equal(grid[w], og) and np.array_equal(avat[w], oa) and np.array_equal(glob[w], ogl)
      want = o.render_world() if kind == E.OBS_WORLD_RGB else np.stack(
          [o.render_agent(p) for p in range(o.P)])
      assert np.array_equal(view[w].cpu().numpy(), want)
    assert eng.counters()["world_steps"] == 12 * n
    eng.close()


def test_callers_tensors_from_the_mapped_pool(clean_up_pack):
  """`memory.mapped_allocations()`: a tensor the CALLER allocates (torch's allocator, torch's
  lifetime) lies in memory mapped from scattered 2 MB chunks — mp_bind_output recognises it as
  a view of the library's, the launch writes it, the oracle agrees — and goes back through
  mp_torch_free (the process's retired address space grows by it)."""
  import gc
  import torch
  from meltingpot_amd import engine as E, memory
  n, T = 700, 2     # RGB of 700 worlds = 114 MB a slot
  eng = _engine(clean_up_pack, n)
  retired = eng.retired_va["bytes"]
  with memory.mapped_allocations():
    ring = torch.empty((T,) + eng.shapes[E.OBS_RGB][0], dtype=torch.uint8, device=eng.device)
    small = torch.empty(1000, dtype=torch.uint8, device=eng.device)     # (< 32 MB: plain hipMalloc)
  outside = torch.empty(8, device=eng.device)
  assert ring.is_cuda and small.is_cuda and outside.is_cuda
  eng.bind_ring(E.OBS_RGB, ring)
  eng.reset()
  rng = np.random.default_rng(4)
  acts = util.random_actions(rng, 3, n, eng.P, eng.num_actions)
  for s in range(3):
    eng.step(torch.from_numpy(acts[s]).to(eng.device))
  for w in (0, 333, n - 1):
    o = util.make_oracles(clean_up_pack, 1, offset=w)[0]
    o.reset()
    for s in range(3):
      o.step(acts[s, w])
      if s >= 1:     # submissions 2 and 3 are what two slots still hold
        assert np.array_equal(ring[(s + 1) % T, w].cpu().numpy(),
                              np.stack([o.render_agent(p) for p in range(o.P)])), (w, s)
  eng.close()
  nbytes = ring.numel()
  del ring, small
  gc.collect()
  torch.cuda.empty_cache()
  probe = _engine(clean_up_pack, 2)
  # (the pool may keep the segment cached; if it was released, the range was retired)
  assert probe.retired_va["bytes"] in (retired, ) or probe.retired_va["bytes"] >= retired + nbytes
  probe.close()
