"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

GOLDEN = 0x9E3779B97F4A7C15
MASK64 = (1 << 64) - 1


def world_seed(w: int) -> int:
  """Default per-world seed of the engine."""
  return (GOLDEN * (w + 1)) & MASK64


def make_oracles(pack_bytes, n, offset=0, num_players=0):
  from oracle import oracle
  return [oracle.Oracle(pack_bytes, world_seed(offset + w), num_players) for w in range(n)]


def _replay_chunk(job):
  """Worker of replay_parallel (its own process: imports only numpy + the oracle)."""
  import os, sys
  here = os.path.dirname(os.path.abspath(__file__))
  for d in (here, os.path.dirname(here)):
    if d not in sys.path:
      sys.path.insert(0, d)
  from oracle import oracle
  pack_bytes, first, acts, looks, sample, world_view, num_players = job
  steps, count = acts.shape[0], acts.shape[1]
  out = []
  for i in range(count):
    w = first + i
    o = oracle.Oracle(pack_bytes, world_seed(w), num_players)
    o.reset()
    views = {}
    for s in range(steps):
      o.step(acts[s, i])
      if w in sample and s + 1 in looks:
        if world_view == "both":
          views[s + 1] = (o.render_world(), np.stack([o.render_agent(p) for p in range(o.P)]))
        else:
          views[s + 1] = (o.render_world() if world_view else
                          np.stack([o.render_agent(p) for p in range(o.P)]))
    g, a, gl = o.dump()
    out.append((w, g, a, gl, o.rewards(), o.events(), views))
    o.close()
  return out


def replay_parallel(pack_bytes, acts, looks=(), sample=(), world_view=True, offset=0,
                    num_players=0, workers=None, timeout=1200):
  """Replays worlds [offset, offset + n) in the oracle for acts [steps, n, P], spread
  over the host's cores (the oracle is scalar C): plain subprocesses of this file
  (`python tests/util.py --replay job.pkl`), jobs and results through pickles in a
  temporary directory.  Yields, per world in order: (w, grid, avat, glob, rewards,
  events, {step: view}) with the benchmarked view of the `sample` worlds after
  the steps in `looks`."""
  import os
  import pickle
  import subprocess
  import sys
  import tempfile
  n = acts.shape[1]
  workers = workers or max(1, min(len(os.sched_getaffinity(0)), 48, (n + 15) // 16))
  chunk = (n + workers - 1) // workers
  jobs = [(pack_bytes, offset + i, np.ascontiguousarray(acts[:, i:i + chunk]), tuple(looks),
           frozenset(sample), world_view, num_players) for i in range(0, n, chunk)]
  if len(jobs) == 1:
    yield from _replay_chunk(jobs[0])
    return
  with tempfile.TemporaryDirectory(prefix="mp_replay_") as tmp:
    procs = []
    for k, job in enumerate(jobs):
      path = os.path.join(tmp, f"job{k}.pkl")
      with open(path, "wb") as f:
        pickle.dump(job, f)
      procs.append((path, subprocess.Popen([sys.executable, os.path.abspath(__file__),
                                            "--replay", path])))
    try:
      for path, pr in procs:
        assert pr.wait(timeout=timeout) == 0, "oracle replay worker failed"
        with open(path + ".out", "rb") as f:
          yield from pickle.load(f)
    finally:
      for _, pr in procs:
        if pr.poll() is None:
          pr.kill()


def pack_tables(pack_bytes):
  from meltingpot_amd import pack
  return pack.loads(pack_bytes)


def random_actions(rng, steps, n, p, nact, weights=None):
  if weights is None:
    return rng.integers(0, nact, size=(steps, n, p), dtype=np.int32)
  w = np.asarray(weights, np.float64)
  return rng.choice(nact, size=(steps, n, p), p=w / w.sum()).astype(np.int32)


# 16 equally likely slots -> action id of externality_mushrooms' ACTION_SET: walks forward, zaps often
ZAP_HEAVY_SLOTS = np.array([0, 1, 1, 1, 1, 2, 3, 4, 5, 5, 6, 6, 7, 7, 7, 7], np.int32)


def hashed_actions(worlds, step, players, slots=ZAP_HEAVY_SLOTS, num_actions=0):
  """int32 [len(worlds), players]: actions as a pure function of (GLOBAL world index, step,
  player) — a splitmix64-style hash picking one of `slots` (or, with `num_actions`, an id below
  it) — so that a world found in a run of thousands (tools/gpu_find_displaced_markings.py,
  tests/tools/deep_soak.py) can be replayed alone, on the oracle and on an engine created with
  world_offset = that world."""
  with np.errstate(over="ignore"):
    w = np.asarray(worlds, np.uint64)[:, None]
    p = np.arange(players, dtype=np.uint64)[None, :]
    x = (w * np.uint64(0x9E3779B97F4A7C15) +
         np.full((1, 1), step, np.uint64) * np.uint64(0xBF58476D1CE4E5B9) +
         p * np.uint64(0x94D049BB133111EB))
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
  if num_actions:
    return ((x >> np.uint64(33)) % np.uint64(num_actions)).astype(np.int32)
  assert len(slots) == 16
  return np.asarray(slots, np.int32)[(x >> np.uint64(60)).astype(np.int64)]


def patch_pack(pack_bytes, tables=None, **hdr_overrides):
  """Returns a pack with some header fields (e.g. MAXFRAMES) or whole tables
  replaced."""
  from meltingpot_amd import lower, pack
  t = pack.loads(pack_bytes)
  for k, v in hdr_overrides.items():
    t["hdr"][getattr(lower, "HDR_" + k)] = v
  for k, v in (tables or {}).items():
    t[k] = np.asarray(v, t[k].dtype).reshape(t[k].shape)
  return pack.dumps(t)


def fertile_clean_up(pack_bytes, depletion=1.0, restoration=0.0, max_rate=0.3,
                     dirt_prob=0.1, dirt_delay=5):
  """clean_up with AppleGrow thresholds moved so that apples grow from the
  first step (random play never cleans the river below thresholdDepletion
  = 0.4, clean_up.py:398-404, so the stock pack exercises no apple code)."""
  from meltingpot_amd import lower, pack
  t = pack.loads(pack_bytes)
  f = t["cu_f64"].copy()
  f[0], f[1], f[2], f[3] = max_rate, depletion, restoration, dirt_prob
  i = t["cu_i32"].copy()
  i[3] = dirt_delay
  thr = lower.clean_up_apple_thresholds(len(t["dirt_cells"]), max_rate,
                                        depletion, restoration)
  misc = t["thr_misc"].copy()
  misc[0] = lower.prob_threshold(dirt_prob)
  return patch_pack(pack_bytes, tables={"cu_f64": f, "cu_i32": i,
                                        "apple_thr": thr, "thr_misc": misc})


def matrix_variant(pack_bytes, *, taste=None, itaste=None, multiplier=None, unready=None,
                   zero_inventory=None, random_tie=None, floor=None, regen_rate=None):
  """An *_in_the_matrix pack with rule constants the stock configs leave at their
  defaults (the reference's scenarios and config overrides set them):
  taste = per player (mostTastyResourceClass, mostTastyReward, defaultTastinessReward)
  (Taste, the_matrix/components.lua:966-990); itaste = per player (class,
  zeroDefaultInteractionReward, extraReward) (InteractionTaste, :993-1039);
  multiplier / unready / floor: rewardMultiplier, rewardFromZappingUnreadyPlayer,
  rewardFloor (:361-376); zero_inventory, random_tie (TheMatrix, :199-206)."""
  from meltingpot_amd import lower, pack
  t = pack.loads(pack_bytes)
  pi = t["mx_player_i32"].reshape(-1, 4).copy()
  pf = t["mx_player_f64"].reshape(-1, 4).copy()
  mi, mf, thr = t["mx_i32"].copy(), t["mx_f64"].copy(), t["mx_thr"].copy()
  for p in range(len(pi)):
    if taste is not None:
      c, r, d = taste[p % len(taste)]
      pi[p, 0] = c; pf[p, 0] = r; pf[p, 1] = d
    if itaste is not None:
      c, z, e = itaste[p % len(itaste)]
      pi[p, 1] = c; pi[p, 2] = int(z); pf[p, 2] = e
  if multiplier is not None: mf[1] = multiplier
  if unready is not None: mf[2] = unready
  if floor is not None: mf[0] = floor
  if zero_inventory is not None: mi[11] = int(zero_inventory)
  if random_tie is not None: mi[12] = int(random_tie)
  if regen_rate is not None:
    mf[3] = regen_rate; thr[0] = lower.prob_threshold(regen_rate)
  return patch_pack(pack_bytes, tables={"mx_player_i32": pi.reshape(t["mx_player_i32"].shape),
                                        "mx_player_f64": pf.reshape(t["mx_player_f64"].shape),
                                        "mx_i32": mi, "mx_f64": mf, "mx_thr": thr})


if __name__ == "__main__":
  import pickle
  import sys
  if len(sys.argv) == 3 and sys.argv[1] == "--replay":   # worker of replay_parallel
    with open(sys.argv[2], "rb") as f:
      job = pickle.load(f)
    with open(sys.argv[2] + ".out", "wb") as f:
      pickle.dump(_replay_chunk(job), f)
