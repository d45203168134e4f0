"""The N > 1 path, exercised with 2 gloo ranks on CPU: world sharding by global
index and the measurement-window reduction that bench.py performs over RCCL."""
import os
import subprocess
import sys
import textwrap

import pytest

import util
from meltingpot_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_partition_the_worlds_and_their_seeds():
  for total, g in [(4096, 8), (4096, 1), (10, 4), (7, 8), (32768, 8)]:
    seen = []
    for r in range(g):
      off, n = sharding.shard(total, r, g)
      seen.extend(range(off, off + n))
    assert seen == list(range(total))
  assert sharding.world_seed(5) == util.world_seed(5)
  assert sharding.world_seed(5, base_seed=100) == 105
  with pytest.raises(ValueError):
    sharding.shard(8, 8, 8)


_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from meltingpot_amd import sharding, engine
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    off, n = sharding.shard(1001, rank, world)
    counters = {k: (rank + 1) * (i + 1) for i, k in enumerate(engine.COUNTER_NAMES)}
    counters["world_steps"] = n
    secs, tot = sharding.reduce_window(1.0 + rank, counters, engine.COUNTER_NAMES, dist)
    assert secs == float(world), secs
    assert tot["world_steps"] == 1001, tot
    assert tot["agent_steps"] == 2 * sum(r + 1 for r in range(world)), tot
    seeds = [sharding.world_seed(off + i) for i in range(n)]
    gathered = [None] * world
    dist.all_gather_object(gathered, seeds)
    flat = [s for part in gathered for s in part]
    assert flat == [sharding.world_seed(i) for i in range(1001)]
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ok_%%d" %% rank), "w").write("ok")
""")


def test_two_rank_gloo_window_reduction(tmp_path):
  script = tmp_path / "worker.py"
  script.write_text(_WORKER % ROOT)
  env = dict(os.environ, MASTER_ADDR="127.0.0.1")
  out = subprocess.run(
      [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
       "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
       "29517", str(script)],
      capture_output=True, text=True, env=env, timeout=240)
  assert out.returncode == 0, out.stdout + out.stderr
  assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


def test_bench_launches_its_own_ranks():
  """`python bench.py --gpus 2` with no launcher around it starts two ranks
  itself (bench.py:_launch_ranks).  Here without a GPU: `--rendezvous-only`
  stops after the ranks have met over gloo and taken their shards; the engine
  run of the same command line is tests/test_gpu_surface.py::
  test_two_ranks_two_engines_through_bench[self]."""
  import json
  env = dict(os.environ)
  for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--worlds", "4096", "--rendezvous-only"],
                       capture_output=True, text=True, env=env, timeout=240, cwd=ROOT)
  assert out.returncode == 0, out.stdout + out.stderr
  line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
  assert line["n_gpus"] == 2 and line["value"] is None
  assert line["ranks"]["count"] == 2 and len(set(line["ranks"]["devices"])) == 2
  assert line["shards"] == [[0, 4096], [4096, 4096]]


def test_eight_ranks_meet_and_shard_32768_worlds():
  """BASELINE.json configs[4] without the hardware: `bench.py --gpus 8 --rendezvous-only`
  starts EIGHT ranks (gloo), every one answers the all-reduce (ranks.count == 8, eight
  different processes), the shards are 8 x 4096 of 32768 worlds in rank order, and the
  seeds each rank would give its first and last world are those of the GLOBAL world
  indices — what makes a sharded run's worlds the single-GPU run's worlds."""
  import json
  env = dict(os.environ)
  for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8",
                        "--worlds", "4096", "--rendezvous-only"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
  assert out.returncode == 0, out.stdout + out.stderr
  line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
  assert line["n_gpus"] == 8 and line["ranks"]["count"] == 8
  assert len(set(line["ranks"]["devices"])) == 8
  assert line["shards"] == [[r * 4096, 4096] for r in range(8)]
  assert line["shard_seeds"] == [[util.world_seed(r * 4096), util.world_seed(r * 4096 + 4095)]
                                 for r in range(8)]
  # unequal shards: 32771 worlds over 8 ranks
  sizes = [sharding.shard(32771, r, 8) for r in range(8)]
  assert sum(n for _, n in sizes) == 32771 and [o for o, _ in sizes] == sorted(o for o, _ in sizes)
  assert max(n for _, n in sizes) - min(n for _, n in sizes) == 1


def test_a_line_with_missing_or_duplicate_ranks_is_refused():
  """bench.py exits non-zero rather than print a line whose `n_gpus` its ranks do not back."""
  sys.path.insert(0, ROOT)
  import bench
  ok = {"count": 4, "backend": "nccl", "devices": [f"cuda:{i}" for i in range(4)], "ms_per_step": [0] * 4}
  assert bench.check_ranks(ok, 4) is None
  assert "3 rank(s)" in bench.check_ranks(dict(ok, count=3), 4)
  assert "same device" in bench.check_ranks(dict(ok, devices=["cuda:0", "cuda:0", "cuda:1", "cuda:2"]), 4)
  assert bench.check_ranks(dict(ok, devices=["cuda:0"] * 4), 4, one_device=True) is None
  assert bench.check_ranks(dict(ok, backend="gloo", devices=["cuda:0"] * 4), 4) is None


def test_bench_refuses_a_rank_count_it_cannot_honour():
  env = dict(os.environ, WORLD_SIZE="4", RANK="0")
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                       capture_output=True, text=True, env=env, timeout=120, cwd=ROOT)
  assert out.returncode != 0 and "WORLD_SIZE=4" in out.stderr
  env.pop("WORLD_SIZE"); env.pop("RANK")
  import torch
  if not torch.cuda.is_available():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"],
                         capture_output=True, text=True, env=env, timeout=120, cwd=ROOT)
    assert out.returncode != 0 and "0 GPU(s) visible" in out.stderr
