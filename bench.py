#!/usr/bin/env python3
"""Headline benchmark: agent-steps/s of the clean_up step + render hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--worlds 4096]
                  [--obs world|agents]

Workload (BASELINE.json configs[1]): clean_up, 7 players, 4096 worlds per GPU,
random actions, observation set {WORLD.RGB} rendered every step into a
device-resident tensor.  One "step" = one mp_step (step kernel) + one render
launch over all worlds of the rank.  Actions are pre-generated on device
(off the clock); inputs are resident in HBM when the timed region starts.
For N > 1 the driver launches one rank per GPU (torch.distributed.run); worlds
are sharded by global index with no data-path collective (weak scaling); RCCL
only reduces the window's wall time (MAX) and the throughput counters (SUM).

The JSON line also carries `roofline` for the dominant kernel (the renderer:
algorithmic bytes = observation bytes written + world records read, per
launch, over the launch's average duration measured with events on the
engine's stream) and `cpu_baseline` (the CPU oracle on a bounded sample).
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
  pack, obs, nact, first, nworlds, budget_steps, budget_s = job
  worlds = [oracle.Oracle(pack, sharding.world_seed(first + w)) for w in range(nworlds)]
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
                 budget_s=20.0):
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
           f"{budget_steps},{budget_s}"]
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
    results = [_cpu_worker((pack, obs, nact, 0, worlds_per_core, budget_steps, budget_s))]
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


def main():
  if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":   # see cpu_baseline
    from meltingpot_amd import engine as E
    sub, obs, nact, first, nworlds, steps, budget = sys.argv[2].split(",")
    print(json.dumps(_cpu_worker((E.load_pack(sub), obs, int(nact), int(first), int(nworlds),
                                  int(steps), float(budget)))))
    return
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--worlds", type=int, default=4096, help="worlds per GPU")
  ap.add_argument("--obs", choices=("world", "agents"), default="world")
  ap.add_argument("--substrate", default="clean_up",
                  choices=("clean_up", "commons_harvest__open", "territory__rooms",
                           "commons_harvest__closed", "commons_harvest__partnership",
                           "territory__open", "territory__inside_out", "coins"))
  ap.add_argument("--beam-skew", type=float, default=0.0,
                  help="fraction of actions replaced by the substrate's two "
                       "beam actions (SURVEY 8d config 4 uses 0.5)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--host-actions", action="store_true",
                  help="hand the actions over as host arrays (mp_step_host): the "
                       "PCIe-inclusive rate noted in DESIGN.md, never the headline value")
  args = ap.parse_args()

  import torch
  from meltingpot_amd import engine as E
  from meltingpot_amd import sharding

  world_size = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  dist = None
  if world_size > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
  dev = local_rank
  torch.cuda.set_device(dev)

  pack = E.load_pack(args.substrate)
  N = args.worlds  # per GPU: weak scaling
  offset, _ = sharding.shard(N * world_size, rank, world_size)
  eng = E.Engine(pack, N, device=dev, auto_reset=True, world_offset=offset)
  P = eng.P
  kind = E.OBS_WORLD_RGB if args.obs == "world" else E.OBS_RGB
  obs = eng.empty(kind)
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
  eng.reset()
  for i in range(Wm):
    eng.step(acts[i % T])
    eng.observe(kind, obs)

  mk = lambda: torch.cuda.Event(enable_timing=True)
  ev = [(mk(), mk(), mk()) for _ in range(K)]
  if dist is not None:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(K):
    e0, e1, e2 = ev[i]
    e0.record()
    eng.step(acts[(Wm + i) % T])
    e1.record()
    eng.observe(kind, obs)
    e2.record()
  torch.cuda.synchronize()
  if dist is not None:
    dist.barrier()
  dt = time.perf_counter() - t0
  step_all = sorted(a.elapsed_time(b) for a, b, _ in ev)
  render_all = sorted(b.elapsed_time(c) for _, b, c in ev)
  step_ms = sum(step_all) / K
  render_ms = sum(render_all) / K

  dt, counters = sharding.reduce_window(dt, eng.counters(), E.COUNTER_NAMES, dist,
                                        eng.device)

  if rank == 0:
    info = eng.info
    obs_name = "WORLD.RGB" if args.obs == "world" else f"N.RGB x{P}"
    obs_bytes = obs.numel() // N           # per world-step
    state_bytes = info.world_state_bytes   # read once by the render kernel
    alg_bytes = (obs_bytes + state_bytes) * N   # per render launch
    achieved = alg_bytes / (render_ms * 1e-3) / 1e9
    workload = (f"{args.substrate}, {P} players, {N} worlds/GPU, random actions"
                + (" handed over as host arrays (PCIe-inclusive)" if args.host_actions else "")
                + (f" ({args.beam_skew:.0%} beam actions)" if args.beam_skew > 0 else "")
                + f", obs={{{obs_name}}} rendered every step")
    if args.obs == "world" and args.substrate == "clean_up":
      workload += " (BASELINE.json configs[1])"
    if args.obs == "agents" and args.substrate == "commons_harvest__open":
      workload += " (BASELINE.json configs[2])"
    if args.obs == "agents" and args.substrate == "territory__rooms":
      workload += " (BASELINE.json configs[3])"
    traffic = None
    try:  # HBM bytes per launch from the committed PMC profile of this very config
      with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
        traffic = json.load(f).get(f"{args.substrate}/{N}/{args.obs}", {}).get("k_render")
    except (OSError, ValueError):
      pass
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
            "bound": "hbm",
            "kernel": "k_render<%s>" % ("world" if args.obs == "world" else "agents"),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "bytes_per_launch": alg_bytes, "avg_launch_ms": render_ms,
        },
        "kernels_ms": {"step": step_ms, "render": render_ms,
                       "render_min": render_all[0], "render_median": render_all[K // 2],
                       "render_max": render_all[-1], "step_min": step_all[0]},
        "counters": counters,
        "cpu_baseline": None,
    }
    if world_size == 1 and not args.no_cpu_baseline:
      line["cpu_baseline"] = cpu_baseline(args.substrate, pack, args.obs, eng.num_actions)
    print(json.dumps(line))
  eng.close()
  if dist is not None:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
