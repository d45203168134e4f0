#!/usr/bin/env python3
"""Headline benchmark: agent-steps/s of the clean_up step + render hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--worlds 4096]
                  [--obs world|agents] [--unfused | --fused]

Workload (BASELINE.json configs[1]): clean_up, 7 players, 4096 worlds per GPU,
random actions, observation set {WORLD.RGB} rendered every step into a
device-resident tensor bound to the engine.  One "step" = one mp_step: ONE
persistent launch (k_frame) that steps every world of the rank and renders the
bound view, or (`--unfused` only; the engine itself always fuses) one launch for
the rules and one for the pixels — the roofline object then describes the
second, the dominant one, and `kernels_ms` carries both.  Actions are pre-generated on device (off the clock);
inputs are resident in HBM when the timed region starts.  For N > 1 there is one
rank per GPU: either the caller launches them (torch.distributed.run, the
driver's form) or, when `--gpus N` is given and WORLD_SIZE is not set, this
script launches them itself (the same torch.distributed.run command line);
worlds are sharded by global index with no data-path collective (weak scaling);
RCCL only reduces the window's wall time (MAX), the throughput counters (SUM)
and the rank evidence of the JSON line (`ranks`: an all-reduced count of ones,
every rank's device and its own ms_per_step).

The JSON line also carries `roofline` for the dominant kernel: algorithmic bytes
per launch (observation bytes written + records read and written + actions +
scalar outputs) over the launch's average duration, from one pair of events on
the engine's stream around the timed region; `traffic` = HBM bytes per launch
from two rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950
correction of MI355X_MICROARCH.md) that this script runs on itself after the
timed region; `cpu_baseline` (the CPU oracle on a bounded sample); and, next to the
headline, `substrate_api`: the same 4096 worlds behind the drop-in surface a
training loop binds — `substrate.build("clean_up", roles=("default",) * 7,
num_worlds=4096)`, i.e. per-agent RGB AND WORLD.RGB plus six scalar kinds — stepped
with device actions; one fused launch draws both views.  An extra key, never
part of `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def _cpu_worker(job):
  """One host core: `nworlds` oracle worlds stepped and rendered for up to
  `budget_steps` steps / `budget_s` seconds.  Returns (agent-steps, seconds)."""
  import numpy as np
  from meltingpot_amd import sharding
  from oracle import oracle  # the checker, timed as the reported CPU baseline
  pack, obs, nact, first, nworlds, budget_steps, budget_s, players = job
  worlds = [oracle.Oracle(pack, sharding.world_seed(first + w), players) for w in range(nworlds)]
  for o in worlds:
    o.reset()
  P = worlds[0].P
  rng = np.random.default_rng(1234 + first)
  acts = rng.integers(0, nact, size=(budget_steps, nworlds, P), dtype=np.int32)
  t0 = time.perf_counter()
  done_steps = 0
  for s in range(budget_steps):
    for w, o in enumerate(worlds):
      o.step(acts[s, w])
      if obs == "world":
        o.render_world()
      else:
        for p in range(P):
          o.render_agent(p)
    done_steps += 1
    if time.perf_counter() - t0 > budget_s:
      break
  return nworlds * P * done_steps, time.perf_counter() - t0, done_steps, P


def cpu_baseline(substrate, pack, obs, nact, worlds_per_core=16, budget_steps=1000,
                 budget_s=20.0, players=0):
  """Times the CPU oracle (scalar C restatement, one process per host core, worlds
  sharded over the cores like the reference would run one DMLab2D per core) on a
  bounded sample of the same workload, same observation set rendered every step.
  Workers are plain subprocesses of this script (`--cpu-worker`); a worker that
  fails or overruns is dropped, and with none left the sample runs in-process."""
  import subprocess
  cores = max(1, min(len(os.sched_getaffinity(0)), 64))
  procs = []
  for c in range(cores):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker",
           f"{substrate},{obs},{nact},{c * worlds_per_core},{worlds_per_core},"
           f"{budget_steps},{budget_s},{players}"]
    procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                  text=True))
  results = []
  deadline = time.perf_counter() + budget_s + 90.0
  for pr in procs:
    try:
      out, _ = pr.communicate(timeout=max(1.0, deadline - time.perf_counter()))
      if pr.returncode == 0:
        results.append(tuple(json.loads(out.strip().splitlines()[-1])))
    except (subprocess.TimeoutExpired, ValueError, IndexError):
      pr.kill()
  if not results:
    results = [_cpu_worker((pack, obs, nact, 0, worlds_per_core, budget_steps, budget_s, players))]
  total = sum(r[0] for r in results)
  dt = max(r[1] for r in results)
  steps, P = results[0][2], results[0][3]
  return {
      "value": total / dt,
      "unit": "agent-steps/s",
      "cores": len(results),
      "kind": "port",
      "sample": f"{len(results)} cores x {worlds_per_core} worlds x ~{steps} steps, {P} players, "
                f"obs={'WORLD.RGB' if obs == 'world' else 'per-agent RGB'}, "
                f"oracle/liboracle.so (gcc -O3, one process per core), {dt:.1f} s",
  }


def _measure_traffic(argv, kernel_substr, timeout_s=120):
  """HBM bytes per launch of the kernels matching `kernel_substr`: two
  `rocprofv3 --pmc` passes (one counter each, MI355X_MICROARCH.md) over a short
  child run of this script with the same workload flags.  None if rocprofv3 is
  missing or a pass fails."""
  import glob
  import shutil
  import sqlite3
  import subprocess
  import tempfile
  rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
  if not os.path.exists(rocprof):
    return None
  vals = {}
  tmp = tempfile.mkdtemp(prefix="mp_pmc_", dir="/tmp")
  env = dict(os.environ, TMPDIR="/tmp")
  try:
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
      out = os.path.join(tmp, counter)
      cmd = [rocprof, "--pmc", counter, "-d", out, "-o", "r", "--", sys.executable,
             os.path.abspath(__file__), "--steps", "12", "--warmup", "2", "--no-cpu-baseline",
             "--no-traffic", "--no-substrate-api", "--no-rollout-api", "--no-steady-state",
             "--no-configs", "--no-box-fill"] + argv
      try:
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, check=True)
      except (subprocess.SubprocessError, OSError):
        return None
      dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
      if not dbs:
        return None
      rows = sqlite3.connect(dbs[0]).execute(
          "select kernel_name, avg(value) from counters_collection where counter_name = ? "
          "group by kernel_name", (counter,)).fetchall()
      hit = [v for k, v in rows if kernel_substr in k]
      if not hit:
        return None
      vals[counter] = max(hit)   # KiB per dispatch
    return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
  finally:
    shutil.rmtree(tmp, ignore_errors=True)


def algorithmic_bytes(info, obs_bytes_per_world, P, N):
  """Algorithmic HBM bytes of ONE fused launch: the observation written + the
  world records read and written + actions (i32 per player) + per-player outputs (reward,
  ready, metric f64; position 2 x i32, orientation i32) + per-world outputs (collective f64,
  step type i32, discount f64, events header row 16 B) [+ *_in_the_matrix: INVENTORY [P][R]
  and INTERACTION_INVENTORIES [P][2][R], f64]."""
  scalar_bytes = 4 * P + (3 * 8 + 12) * P + 36 + 3 * 8 * P * info.num_resources
  return (obs_bytes_per_world + 2 * info.world_state_bytes + scalar_bytes) * N


def with_box_fill(fills, alg_bytes, launch_ms):
  """`box_fill` of a leg: mp_box_fill of every pixel view the launch writes, on the SAME bound
  buffers (memset / bare store loop in the product's order / the same as a chip-wide 4 KiB
  front, us each), and `frac_of_box_fill` = the launch's algorithmic byte rate over the rate at
  which this box fills these buffers in the best of the three ways: what is left of the figure
  once the box and the buffers are taken out of it."""
  best_us = sum(min(f["memset_us"], f["product_order_us"], f["front_4k_us"]) for f in fills.values())
  view_bytes = sum(f["bytes"] for f in fills.values())
  out = {"views": fills, "best_fill_us": round(best_us, 2),
         "best_fill_GBs": round(view_bytes / best_us / 1e3, 1)}
  out["frac_of_box_fill"] = (alg_bytes / (launch_ms * 1e3)) / (view_bytes / best_us)
  return out


def config_leg(substrate, players, num_worlds, beam_skew, steps, warmup, device, place, config_name):
  """One more single-GPU BASELINE.json config next to the headline (an extra key of the line,
  never `value`): per-agent RGB of every player, one fused launch per step, its own engine,
  placement probe and tuned plan — measured like the headline (events on the engine's stream
  around `steps` launches after `warmup`), plus the box calibration on its bound buffer."""
  import torch
  from meltingpot_amd import engine as E
  pack = E.load_pack(substrate)
  eng = E.Engine(pack, num_worlds, device=device, auto_reset=True, num_players=players,
                 placements=place)
  N, P = eng.N, eng.P
  gen = torch.Generator(device=eng.device)
  gen.manual_seed(99)
  T = 64
  acts = torch.randint(0, eng.num_actions, (T, N, P), generator=gen, device=eng.device,
                       dtype=torch.int32)
  if beam_skew > 0:
    na = eng.num_actions
    beam = torch.randint(na - 2, na, (T, N, P), generator=gen, device=eng.device, dtype=torch.int32)
    pick = torch.rand((T, N, P), generator=gen, device=eng.device) < beam_skew
    acts = torch.where(pick, beam, acts)
  t0 = time.perf_counter()
  obs = eng.bind(E.OBS_RGB)
  bind_s = time.perf_counter() - t0
  eng.reset()
  for i in range(warmup):
    eng.step(acts[i % T])
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  e0.record()
  for i in range(steps):
    eng.step(acts[(warmup + i) % T])
  e1.record()
  while not e1.query():
    pass
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  launch_ms = e0.elapsed_time(e1) / steps
  alg = algorithmic_bytes(eng.info, obs.numel() // N, P, N)
  out = {
      "workload": f"{substrate}, {P} players, {N} worlds, random actions"
                  + (f" ({beam_skew:.0%} beam actions)" if beam_skew > 0 else "")
                  + f", obs={{N.RGB x{P}}} rendered every step, one fused launch per step "
                  + f"(BASELINE.json {config_name})",
      "value": N * P * steps / dt, "unit": "agent-steps/s", "steps": steps, "warmup": warmup,
      "ms_per_step": dt / steps * 1e3, "avg_launch_ms": launch_ms,
      "launches_per_step": 1 if eng.fused else 2,
      "bytes_per_launch": alg, "achieved": alg / (launch_ms * 1e-3) / 1e9,
      "frac": alg / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
      "placement": dict(eng.placement.get(E.OBS_RGB) or {}, bind_s=round(bind_s, 3)),
      "plan": eng.plan,
      "counters": eng.counters(),
  }
  out["box_fill"] = with_box_fill({"RGB": eng.box_fill(E.OBS_RGB)}, alg, launch_ms)
  eng.close()
  del obs
  return out


def substrate_api_bench(num_worlds, steps, warmup, device):
  """`substrate.build("clean_up", roles=("default",) * 7, num_worlds=N)` (the
  reference's meltingpot.substrate.build + `num_worlds`: utils/substrates/
  substrate.py:66-81, configs/substrates/clean_up.py:813-832) stepped `steps` times
  with device-resident actions: every TimeStep leaf (RGB, WORLD.RGB, READY_TO_SHOOT,
  NUM_OTHERS_WHO_CLEANED_THIS_STEP, COLLECTIVE_REWARD, reward, discount, step_type)
  is a bound device tensor refreshed by the step's ONE launch."""
  import torch
  from meltingpot_amd import engine as E
  from meltingpot_amd import substrate
  env = substrate.build("clean_up", roles=("default",) * 7, num_worlds=num_worlds, device=device)
  eng = env.engine
  N, P = eng.N, eng.P
  gen = torch.Generator(device=eng.device)
  gen.manual_seed(4321)
  T = min(steps + warmup, 128)
  acts = torch.randint(0, eng.num_actions, (T, N, P), generator=gen, device=eng.device,
                       dtype=torch.int32)
  env.reset()
  for i in range(warmup):
    env.step(acts[i % T])
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  e0.record()
  for i in range(steps):
    ts = env.step(acts[(warmup + i) % T])
  e1.record()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  launch_ms = e0.elapsed_time(e1) / steps
  info = eng.info
  rgb_bytes = ts.observation["RGB"].numel() // N
  wrgb_bytes = ts.observation["WORLD.RGB"].numel() // N
  scalar_bytes = 4 * P + (3 * 8 + 12) * P + 36
  alg = (rgb_bytes + wrgb_bytes + 2 * info.world_state_bytes + scalar_bytes) * N
  out = {
      "api": 'substrate.build("clean_up", roles=("default",) * 7, num_worlds=%d): per-agent RGB + '
             "WORLD.RGB + six scalar kinds bound, device actions" % N,
      "value": N * P * steps / dt, "unit": "agent-steps/s", "steps": steps, "warmup": warmup,
      "ms_per_step": dt / steps * 1e3, "avg_launch_ms": launch_ms,
      "launches_per_step": 1 if eng.fused else 3,
      "bytes_per_launch": alg, "achieved": alg / (launch_ms * 1e-3) / 1e9,
      "frac": alg / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
      "placement": {("RGB" if k == E.OBS_RGB else "WORLD.RGB"): v for k, v in eng.placement.items()},
      "plan": eng.plan,
  }
  out["box_fill"] = with_box_fill({"RGB": eng.box_fill(E.OBS_RGB),
                                   "WORLD.RGB": eng.box_fill(E.OBS_WORLD_RGB)}, alg, launch_ms)
  env.close()
  return out


def rollout_api_bench(num_worlds, steps, warmup, device, slots=32):
  """Observations a learner KEEPS, three ways, same 4096 clean_up worlds and per-agent RGB
  (0.67 GB a step): (1) `single`: one bound buffer overwritten in place — what the headline
  form measures, nothing is kept; (2) `clone`: the same, plus the copy of every step's
  observation into a [T, N, ...] rollout buffer that a PPO-style loop needs to keep it
  (`rollout[t % T].copy_(obs)`: the reference hands back fresh arrays, utils/substrates/
  substrate.py:74-81, so a port of such a loop to an in-place buffer has to clone); (3)
  `ring`: mp_bind_output_ring — step t's launch writes slot t % T of that buffer itself,
  zero copies, no synchronisation, a launch plan per slot tuned once at bind time."""
  import torch
  from meltingpot_amd import engine as E
  pack = E.load_pack("clean_up")
  out = {"slots": slots, "worlds": num_worlds, "view": "per-agent RGB", "steps": steps,
         "warmup": warmup, "unit": "agent-steps/s"}
  gen = torch.Generator(device=f"cuda:{device}")
  gen.manual_seed(77)

  def timed(eng, after_step=None):
    N, P = eng.N, eng.P
    acts = torch.randint(0, eng.num_actions, (64, N, P), generator=gen, device=eng.device,
                         dtype=torch.int32)
    eng.reset()
    for i in range(warmup):
      eng.step(acts[i % 64])
      if after_step:
        after_step(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
      eng.step(acts[(warmup + i) % 64])
      if after_step:
        after_step(warmup + i)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": N * P * steps / dt, "ms_per_step": dt / steps * 1e3,
            "events_ms_per_step": e0.elapsed_time(e1) / steps}

  eng = E.Engine(pack, num_worlds, device=device)
  t0 = time.perf_counter()
  obs = eng.bind(E.OBS_RGB)                       # placed: the fastest of the probe's candidates
  setup_s = round(time.perf_counter() - t0, 2)
  out["single"] = dict(timed(eng), setup_s=setup_s,
                       placement=eng.placement.get(E.OBS_RGB), plan=eng.plan)
  alg = algorithmic_bytes(eng.info, obs.numel() // eng.N, eng.P, eng.N)
  out["single"]["box_fill"] = with_box_fill({"RGB": eng.box_fill(E.OBS_RGB)}, alg,
                                            out["single"]["events_ms_per_step"])
  rollout = eng.empty_ring(E.OBS_RGB, slots)      # [T, N, P, 88, 88, 3]
  out["bytes"] = {"rollout_buffer": rollout.numel(), "per_step": obs.numel()}
  out["clone"] = timed(eng, lambda i: rollout[i % slots].copy_(obs))
  eng.unbind(E.OBS_RGB)
  del obs, rollout
  t0 = time.perf_counter()
  ring = eng.bind_ring(E.OBS_RGB, slots=slots)    # scattered 2 MB chunks, a plan per slot (untimed set-up)
  out["ring"] = dict(timed(eng), setup_s=round(time.perf_counter() - t0, 2))
  # what the single buffer above would be WITHOUT its placement probe (the mean of the probe's
  # candidates): a ring's slots are 32 such draws, the single buffer the best of up to 24
  cand = (out["single"]["placement"] or {}).get("dry_launch_us") or []
  if cand:
    out["ring"]["vs_mean_single_candidate"] = (sum(cand) / len(cand)) / (out["ring"]["events_ms_per_step"] * 1e3)
  del ring
  out["ring_vs_single"] = out["ring"]["value"] / out["single"]["value"]
  out["ring_vs_clone"] = out["ring"]["value"] / out["clone"]["value"]
  eng.close()
  return out


def _free_port():
  import socket
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def _launch_ranks(n, one_device, rendezvous_only):
  """`python bench.py --gpus N` without a launcher: re-runs this command line
  under torch.distributed.run, one rank per GPU of this node (RCCL), and
  returns its exit status.  Refuses more ranks than visible devices."""
  import subprocess
  if not (one_device or rendezvous_only):
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n > have:
      raise SystemExit(f"bench.py: --gpus {n} but {have} GPU(s) visible")
  env = dict(os.environ, MASTER_ADDR="127.0.0.1")
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL on this driver)
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.call(cmd, env=env)


def _rank_evidence(dist, device, backend_device, ms_per_step):
  """What shows that `world_size` ranks really ran: ones summed by the
  collective backend, and every rank's device and own time per step."""
  import torch
  ones = torch.ones(1, dtype=torch.int64, device=backend_device)
  dist.all_reduce(ones, op=dist.ReduceOp.SUM)
  mine = {"rank": dist.get_rank(), "device": device, "ms_per_step": ms_per_step}
  everyone = [None] * dist.get_world_size()
  dist.all_gather_object(everyone, mine)
  return {"count": int(ones.item()), "backend": dist.get_backend(),
          "devices": [e["device"] for e in everyone],
          "ms_per_step": [e["ms_per_step"] for e in everyone]}


def check_ranks(ranks, n_gpus, one_device=False):
  """A line that claims N GPUs must have been produced by N ranks on N different
  devices: None if it was, else what is wrong (the caller exits non-zero — a wrong line
  must not look like a measurement)."""
  if ranks["count"] != n_gpus or len(ranks["devices"]) != n_gpus:
    return f"{ranks['count']} rank(s) answered the all-reduce, {n_gpus} were asked for"
  if ranks["backend"] == "nccl" and not one_device and len(set(ranks["devices"])) != n_gpus:
    return f"two ranks report the same device: {ranks['devices']}"
  return None


def _rendezvous_only(args):
  """`--rendezvous-only` (CPU tests of the N > 1 plumbing; never a bench line):
  the ranks meet over gloo, take their world shards and report them — no engine,
  no timing, `value` is null."""
  import torch.distributed as dist
  from meltingpot_amd import sharding
  world_size = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  shards = [sharding.shard(args.worlds * world_size, r, world_size) for r in range(world_size)]
  ranks = {"count": 1, "backend": None, "devices": ["cpu"], "ms_per_step": [None]}
  # what each rank would seed its engine's first and last world with (MpConfig.world_offset:
  # seeds follow the GLOBAL world index, so results do not depend on the number of ranks)
  off, n = shards[rank]
  mine = [sharding.world_seed(off), sharding.world_seed(off + n - 1)]
  seeds = [mine]
  if world_size > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    ranks = _rank_evidence(dist, f"cpu (pid {os.getpid()})", None, None)
    seeds = [None] * world_size
    dist.all_gather_object(seeds, mine)
    dist.barrier()
    dist.destroy_process_group()
  wrong = check_ranks(ranks, args.gpus) if world_size > 1 else None
  if rank == 0:
    print(json.dumps({"metric": "rendezvous only (no engine, not a measurement)", "value": None,
                      "n_gpus": world_size, "ranks": ranks,
                      "shards": [list(s) for s in shards], "shard_seeds": seeds}))
  if wrong:
    raise SystemExit(f"bench.py: {wrong}")


def box_state(device_index):
  """What the box reports about the GPU this line was measured on — the launch is HBM-bound
  and boxes differ: partition modes, clocks and power as `rocm-smi` gives
  them.  Diagnostic only; any failure leaves the field out."""
  import subprocess
  try:
    out = subprocess.run(["rocm-smi", "-d", str(device_index), "--showcomputepartition",
                          "--showmemorypartition", "--showclocks", "--showpower", "--showperflevel",
                          "--json"], capture_output=True, text=True, timeout=10)
    cards = json.loads(out.stdout)
    card = cards.get(f"card{device_index}") or next(iter(cards.values()))
    keep = ("partition", "sclk", "mclk", "fclk", "socclk", "power", "performance level")
    return {k: v for k, v in card.items() if any(w in k.lower() for w in keep)}
  except Exception:   # (no rocm-smi, no JSON, another layout)
    return None


def main():
  if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":   # see cpu_baseline
    from meltingpot_amd import engine as E
    sub, obs, nact, first, nworlds, steps, budget, players = sys.argv[2].split(",")
    print(json.dumps(_cpu_worker((E.load_pack(sub), obs, int(nact), int(first), int(nworlds),
                                  int(steps), float(budget), int(players)))))
    return
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=100)
  ap.add_argument("--worlds", type=int, default=4096, help="worlds per GPU")
  ap.add_argument("--obs", choices=("world", "agents"), default="world")
  ap.add_argument("--substrate", default="clean_up",
                  help="any committed pack (meltingpot_amd/assets/*.mpk); BASELINE.json: "
                       "clean_up, commons_harvest__open, territory__rooms")
  ap.add_argument("--players", type=int, default=0,
                  help="number of players (0: the pack's default; BASELINE.json: 7 / 16 / 9)")
  ap.add_argument("--beam-skew", type=float, default=0.0,
                  help="fraction of actions replaced by the substrate's two "
                       "beam actions (SURVEY 8d config 4 uses 0.5)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-substrate-api", action="store_true",
                  help="skip the `substrate_api` object (the drop-in surface with both views bound)")
  ap.add_argument("--no-rollout-api", action="store_true",
                  help="skip the `rollout_api` object (observations kept in a ring of 32 slots "
                       "vs cloned into one vs overwritten in place)")
  ap.add_argument("--no-steady-state", action="store_true",
                  help="skip the `steady_state` key (1200 more launches behind the timed region: "
                       "traced runs want the timed region to be the last dispatches)")
  ap.add_argument("--no-box-fill", action="store_true",
                  help="skip the `box_fill` objects (mp_box_fill on the bound buffers: memset, the "
                       "product's write order and a 4 KiB front as bare store loops)")
  ap.add_argument("--no-configs", action="store_true",
                  help="skip the `configs` object (BASELINE.json configs[2] and configs[3] as "
                       "legs of their own next to the headline)")
  ap.add_argument("--no-traffic", action="store_true",
                  help="skip the rocprofv3 PMC passes behind roofline.traffic")
  ap.add_argument("--unfused", action="store_true",
                  help="one launch for the rules and one for the pixels (also gives "
                       "per-kernel timing); default: the engine's choice for the substrate")
  ap.add_argument("--fused", action="store_true",
                  help="one launch per step (rules + pixels)")
  ap.add_argument("--one-device", action="store_true",
                  help="tests: every rank uses device 0 (two engines on one GPU)")
  ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                  help="tests: gloo lets two ranks share one GPU (RCCL refuses duplicates)")
  ap.add_argument("--host-actions", action="store_true",
                  help="hand the actions over as host arrays (mp_step_host): the "
                       "PCIe-inclusive rate noted in DESIGN.md, never the headline value")
  ap.add_argument("--cold", action="store_true",
                  help="tools/ only: between steps, stream 1 GiB through the caches and "
                       "synchronise (what a policy's forward pass does to the engine's "
                       "cached records); per-launch times from events.  Not a bench line")
  ap.add_argument("--place", type=int, default=-1,
                  help="candidates mp_place_output may try for the bound view (-1: the engine's "
                       "default, eight — six views mapped from 2 MB chunks and two plain "
                       "allocations; the view is allocated where the launch writes it fastest; "
                       "1 = the first allocation, its plan tuned).  Reported as `placement`")
  ap.add_argument("--place-max-bytes", type=int, default=0,
                  help="bound on the memory mp_place_output keeps alive while it probes, per "
                       "rank (0: a quarter of the free memory of the rank's device, divided by "
                       "the ranks that share the device)")
  ap.add_argument("--placements", type=int, default=0,
                  help="after the timed region: the same launch with the view bound to this "
                       "many OTHER buffers in turn (60 steps each) — how much of the figure is "
                       "the placement of the output in memory (profiles/r03_buffer_placement.md); "
                       "reported as kernels_ms.frame_by_placement, never part of `value`")
  ap.add_argument("--dev-plan", default="",
                  help="tools/ only: MpDevOptions overrides of the launch plan, e.g. "
                       "batch_worlds=3,feeders=6,waves=16,verbose=1; the JSON line is then "
                       "marked dev_plan and is NOT a bench line")
  ap.add_argument("--rendezvous-only", action="store_true",
                  help="tests: launch / meet / shard / report over gloo without an engine "
                       "(runs without a GPU; prints value null)")
  args = ap.parse_args()

  if args.gpus < 1:
    raise SystemExit("bench.py: --gpus must be >= 1")
  if "WORLD_SIZE" not in os.environ and args.gpus > 1:
    sys.exit(_launch_ranks(args.gpus, args.one_device, args.rendezvous_only))
  if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
    raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE="
                     f"{os.environ.get('WORLD_SIZE')}: launch one rank per GPU")
  if args.rendezvous_only:
    return _rendezvous_only(args)

  # A benchmark must not be steerable from the environment: the engine's
  # developer overrides (another build of the library, launch geometry) are
  # refused here.
  # (the library itself reads no environment variable; MP_ENGINE_LIB is the
  # Python binding's switch for A/B runs of another build)
  bad = [k for k in os.environ if k == "MP_ENGINE_LIB"]
  if bad and not os.environ.get("MP_BENCH_ALLOW_DEV_ENV"):
    raise SystemExit(f"bench.py: developer overrides are set ({', '.join(sorted(bad))}); "
                     "unset them (or set MP_BENCH_ALLOW_DEV_ENV=1 for an A/B run whose "
                     "numbers are not bench lines)")
  dev_plan = {k: int(v) for k, v in (kv.split("=") for kv in args.dev_plan.split(",") if kv)}

  if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # dmabuf IPC: the only kind this driver supports (RCCL's and torch's handles fail with
    # `hipIpcGetMemHandle: invalid argument` under the legacy mode).  Set here as well as in
    # _launch_ranks: the driver starts the ranks itself (torch.distributed.run), and the
    # variable has to be in the environment before the HIP runtime of this process comes up
    # — so before torch is imported.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  import torch
  from meltingpot_amd import engine as E
  from meltingpot_amd import sharding

  world_size = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if args.place < 0:
    args.place = 8
  dist = None
  if world_size > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.one_device:
      local_rank = 0
    torch.cuda.set_device(local_rank)
    if args.dist_backend == "nccl":
      dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
      dist.init_process_group("gloo")
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
  dev = local_rank
  torch.cuda.set_device(dev)

  pack = E.load_pack(args.substrate)
  N = args.worlds  # per GPU: weak scaling
  offset, _ = sharding.shard(N * world_size, rank, world_size)
  eng = E.Engine(pack, N, device=dev, auto_reset=True, world_offset=offset,
                 num_players=args.players,
                 unfused=True if args.unfused else (False if args.fused else None),
                 dev=dev_plan or None, placements=args.place)
  P = eng.P
  kind = E.OBS_WORLD_RGB if args.obs == "world" else E.OBS_RGB
  # the placement probe's memory, per rank: explicit, so that N ranks probing at the same
  # time (one per GPU — or, in the tests, several on one GPU) never add up to the device
  sharing = world_size if args.one_device else 1
  eng.place_max_bytes = args.place_max_bytes or torch.cuda.mem_get_info(dev)[0] // (4 * sharing)
  K, Wm = args.steps, args.warmup
  gen = torch.Generator(device=eng.device)
  gen.manual_seed(1234 + rank)
  T = min(K + Wm, 256)  # action ring, pre-generated off the clock
  acts = torch.randint(0, eng.num_actions, (T, N, P), generator=gen,
                       device=eng.device, dtype=torch.int32)
  if args.beam_skew > 0:
    na = eng.num_actions
    beam = torch.randint(na - 2, na, (T, N, P), generator=gen, device=eng.device,
                         dtype=torch.int32)
    pick = torch.rand((T, N, P), generator=gen, device=eng.device) < args.beam_skew
    acts = torch.where(pick, beam, acts)
  if args.host_actions:   # numpy arrays: Engine.step routes them through mp_step_host
    acts = [a.cpu().numpy() for a in acts]
  # Everything that keeps the host busy comes first; the view is bound (placed:
  # Engine.place() times dry launches on its candidates) right before the warm-up
  # steps.  A GPU that has idled for a few ms runs its next ~150 launches 5 - 20 %
  # slower (clock ramp: tools/gpu_step_series.py), and a short timed region — the
  # driver's --steps 20 --warmup 5 is 2.5 ms — sits entirely in that ramp unless the
  # device was busy just before.
  # (bound BEFORE the first reset: an engine nothing has been done with is really
  # stepped by mp_tune / mp_place_output, behind a copy of its state — a dry launch
  # ranks plans a few per cent apart wrongly)
  t_bind = time.perf_counter()
  obs = eng.bind(kind)     # every step renders the view straight into this tensor
  setup_s = time.perf_counter() - t_bind   # the placement probe + the tuner: paid once
  eng.reset()
  unfused = not eng.fused   # the launch form of a step with this view bound
  plan_used = eng.plan
  for i in range(Wm):
    eng.step(acts[i % T])

  mk = lambda: torch.cuda.Event(enable_timing=True)
  e_begin, e_end = mk(), mk()   # on torch's current stream = the engine's stream
  if dist is not None:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  e_begin.record()
  for i in range(K):
    eng.step(acts[(Wm + i) % T])
  e_end.record()
  # (the host waits for the last step by polling its event, then synchronises: a blocking
  # wait alone adds the interrupt's wake-up latency — tens of us, 2 - 4 % of a 20-step window —
  # to a region whose work is done)
  while not e_end.query():
    pass
  torch.cuda.synchronize()
  if dist is not None:
    dist.barrier()
  dt = time.perf_counter() - t0
  launch_ms = e_begin.elapsed_time(e_end) / K   # GPU time per step, launch gaps included

  kernels_ms = {"frame": launch_ms}
  window_counters = eng.counters()   # (of the warm-up + the timed region: read before anything else steps)
  steady = None
  if (world_size == 1 and not args.cold and not args.host_actions and not dev_plan and
      not args.no_steady_state):
    # Steady state, on record next to the window above (never `value`): the same launch on the
    # same buffer once the device has been busy for a tenth of a second — 1000 more steps, then
    # 200 between two events.  A GPU that idled a few ms runs its next ~150 launches 5 - 20 %
    # slower (clock ramp, profiles/r03_clock_ramp.md); a training loop never lets it idle.
    for i in range(1000):
      eng.step(acts[i % T])
    s0, s1 = mk(), mk()
    s0.record()
    for i in range(200):
      eng.step(acts[i % T])
    s1.record()
    torch.cuda.synchronize()
    steady = {"warmup_steps": Wm + K + 1000, "steps": 200, "avg_launch_ms": s0.elapsed_time(s1) / 200}
  fill = None
  if rank == 0 and world_size == 1 and not unfused and not args.no_box_fill:
    fill = eng.box_fill(kind)   # (overwrites the view with junk: everything below redraws it)
    eng.step(acts[0])
  if args.cold:
    junk = torch.empty(1 << 30, dtype=torch.uint8, device=eng.device)
    cold = []
    for i in range(min(K, 40)):
      junk.fill_(i & 255)
      torch.cuda.synchronize()
      a0, a1 = mk(), mk()
      a0.record(); eng.step(acts[i % T]); a1.record()
      torch.cuda.synchronize()
      cold.append(a0.elapsed_time(a1))
    cold.sort()
    kernels_ms["frame_cold_median"] = cold[len(cold) // 2]
    kernels_ms["frame_cold_min"] = cold[0]
    del junk
  if args.placements > 0 and not unfused and not args.host_actions:
    others = [torch.empty_like(obs) for _ in range(args.placements)]
    by_placement = []
    for buf in others + [obs]:
      eng.bind(kind, buf)
      for i in range(10):
        eng.step(acts[i % T])
      a0, a1 = mk(), mk()
      a0.record()
      for i in range(60):
        eng.step(acts[i % T])
      a1.record()
      torch.cuda.synchronize()
      by_placement.append(a0.elapsed_time(a1) / 60)
    kernels_ms["frame_by_placement"] = by_placement   # the last one: the timed region's own buffer
    del others
  if unfused:
    # per-kernel durations (two launches per step), outside the timed region
    ev = [(mk(), mk(), mk()) for _ in range(min(K, 50))]
    eng.unbind(kind)
    for i, (a0, a1, a2) in enumerate(ev):
      a0.record(); eng.step(acts[i % T]); a1.record(); eng.observe(kind, obs); a2.record()
    torch.cuda.synchronize()
    eng.bind(kind, obs)
    st = sorted(a.elapsed_time(b) for a, b, _ in ev)
    rd = sorted(b.elapsed_time(c) for _, b, c in ev)
    kernels_ms = {"step": sum(st) / len(st), "render": sum(rd) / len(rd),
                  "step_min": st[0], "render_min": rd[0], "render_median": rd[len(rd) // 2],
                  "render_max": rd[-1], "frame": launch_ms}

  backend_device = eng.device if args.dist_backend == "nccl" else None
  local_ms = dt / K * 1e3
  dt, counters = sharding.reduce_window(dt, window_counters, E.COUNTER_NAMES, dist, backend_device)
  ranks = None
  if dist is not None:
    ranks = _rank_evidence(dist, f"cuda:{dev} {torch.cuda.get_device_name(dev)}", backend_device,
                           local_ms)

  wrong = check_ranks(ranks, args.gpus, args.one_device) if ranks is not None else None
  if wrong:
    eng.close()
    raise SystemExit(f"bench.py: {wrong}")
  if rank == 0:
    info = eng.info
    obs_name = "WORLD.RGB" if args.obs == "world" else f"N.RGB x{P}"
    obs_bytes = obs.numel() // N           # per world-step
    state_bytes = info.world_state_bytes   # read once and written once per step
    alg_bytes = algorithmic_bytes(info, obs_bytes, P, N)   # per launch
    if unfused:   # the renderer alone: pixels + the records it reads
      alg_bytes = (obs_bytes + state_bytes) * N
      launch_for_roofline = kernels_ms["render"]
      kernel = "k_frame<render only, %s>" % args.obs
    else:
      launch_for_roofline = launch_ms
      kernel = "k_frame<%s step + render, %s>" % (args.substrate.split("__")[0], args.obs)
    achieved = alg_bytes / (launch_for_roofline * 1e-3) / 1e9
    workload = (f"{args.substrate}, {P} players, {N} worlds/GPU, random actions"
                + (" handed over as host arrays (PCIe-inclusive)" if args.host_actions else "")
                + (f" ({args.beam_skew:.0%} beam actions)" if args.beam_skew > 0 else "")
                + f", obs={{{obs_name}}} rendered every step"
                + (", one launch for the rules + one for the pixels" if unfused else
                   ", one fused launch per step"))
    if args.obs == "world" and args.substrate == "clean_up" and P == 7:
      workload += " (BASELINE.json configs[1])"
    if args.obs == "agents" and args.substrate == "commons_harvest__open" and P == 16:
      workload += " (BASELINE.json configs[2])"
    if args.obs == "agents" and args.substrate == "territory__rooms" and P == 9:
      workload += " (BASELINE.json configs[3])"
    traffic, traffic_source = None, None
    if world_size == 1 and not args.no_traffic:
      child = ["--worlds", str(N), "--obs", args.obs, "--substrate", args.substrate,
               "--players", str(args.players), "--beam-skew", str(args.beam_skew),
               "--unfused" if unfused else "--fused", "--place", "1"]
      traffic = _measure_traffic(child, "k_frame")
      traffic_source = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run" if traffic else None
    line = {
        "metric": ("agent-steps/sec (env_batch x players / wall s), clean_up @4096 worlds"
                   if args.substrate == "clean_up" else
                   f"agent-steps/sec (env_batch x players / wall s), {args.substrate} @{N} worlds"),
        "value": world_size * N * P * K / dt,
        "unit": "agent-steps/s",
        "n_gpus": world_size,
        "steps": K,
        "warmup": Wm,
        "ms_per_step": dt / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": workload, "worlds_per_gpu": N, "players": P,
                   "parallelism": f"worlds sharded over {world_size} GPU(s)"},
        "roofline": {
            "bound": "hbm", "kernel": kernel,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": traffic_source,
            "bytes_per_launch": alg_bytes, "avg_launch_ms": launch_for_roofline,
        },
        "kernels_ms": kernels_ms,
        "counters": counters,
        "cpu_baseline": None,
    }
    if ranks is not None:
      line["ranks"] = ranks
    # where the bound view was allocated: Engine.place()'s dry-launch probe of its
    # candidates (outside the timed region; `value` is measured on the one it kept)
    if steady is not None and not unfused:
      steady["achieved"] = alg_bytes / (steady["avg_launch_ms"] * 1e-3) / 1e9
      steady["frac"] = steady["achieved"] / HBM_PEAK_GBS
      steady["value"] = N * P / (steady["avg_launch_ms"] * 1e-3)
      line["steady_state"] = steady
    if fill is not None:
      line["box_fill"] = with_box_fill({obs_name.split(" ")[0]: fill}, alg_bytes, launch_for_roofline)
    line["placement"] = eng.placement.get(kind)
    if line["placement"] is not None:
      line["placement"]["bind_s"] = round(setup_s, 3)   # wall time of Engine.bind: probe + tuner
    line["plan"] = plan_used   # the launch plan mp_tune kept for this buffer (MpInfo.plan_*)
    box = box_state(int(dev))
    if box:
      line["box"] = box
    if dev_plan:
      line["dev_plan"] = dev_plan   # a tools/ sweep, not a bench line
    if args.cold:
      line["cold_run"] = True       # (extra per-launch timings; not a bench line)
    if world_size == 1 and not args.no_cpu_baseline:
      line["cpu_baseline"] = cpu_baseline(args.substrate, pack, args.obs, eng.num_actions,
                                          players=P)
  want_api = (rank == 0 and world_size == 1 and not args.no_substrate_api and not dev_plan and
              args.substrate == "clean_up" and args.obs == "world" and not args.host_actions)
  want_configs = (rank == 0 and world_size == 1 and not args.no_configs and not dev_plan and
                  args.substrate == "clean_up" and args.obs == "world" and not args.host_actions
                  and not unfused)
  eng.close()
  if rank == 0:
    if want_api:
      del obs                    # (the placed view goes back to the driver with it)
      line["substrate_api"] = substrate_api_bench(N, min(K, 200), min(Wm, 100), dev)
      if not args.no_rollout_api:
        line["rollout_api"] = rollout_api_bench(N, min(K, 200), min(Wm, 100), dev)
    if want_configs:
      # the other two single-GPU configs of BASELINE.json, in steady state (300 warm-up steps:
      # territory's step grows with the claimed area over the first few hundred frames)
      if not want_api:
        del obs
      line["configs"] = {
          "commons_harvest__open": config_leg("commons_harvest__open", 16, 4096, 0.0, max(K, 200),
                                              max(Wm, 300), dev, args.place, "configs[2]"),
          "territory__rooms": config_leg("territory__rooms", 9, 8192, 0.5, max(K, 200),
                                         max(Wm, 300), dev, args.place, "configs[3]"),
      }
    print(json.dumps(line))
  if dist is not None:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
