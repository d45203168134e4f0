"""Multi-GPU layout of the hot path: worlds shard, nothing else does.

Worlds are independent (each reference env is its own Lua VM,
utils/substrates/builder.py:179-187), so G ranks each own a contiguous block of
worlds and step them with no data-path collective.  Per-world seeds derive from
the GLOBAL world index, so results do not depend on G.  The only collectives
are, once per measurement window, a MAX over the ranks' wall time and a SUM
over their throughput counters (RCCL on GPUs — backend "nccl"; gloo in the CPU
tests).
"""

from __future__ import annotations

from typing import Dict, Sequence, Tuple

GOLDEN = 0x9E3779B97F4A7C15
_MASK64 = (1 << 64) - 1


def world_seed(global_world: int, base_seed: int = 0) -> int:
  """Seed of a world (include/mp_engine.h MpConfig.base_seed)."""
  if base_seed:
    return (base_seed + global_world) & _MASK64
  return (GOLDEN * (global_world + 1)) & _MASK64


def shard(total_worlds: int, rank: int, world_size: int) -> Tuple[int, int]:
  """(world_offset, num_worlds) of `rank`: contiguous blocks, sizes differ by
  at most one."""
  if not 0 <= rank < world_size:
    raise ValueError(f"rank {rank} outside [0, {world_size})")
  base, extra = divmod(total_worlds, world_size)
  count = base + (1 if rank < extra else 0)
  offset = rank * base + min(rank, extra)
  return offset, count


def reduce_window(local_seconds: float, local_counters: Dict[str, int],
                  names: Sequence[str], dist=None, device=None):
  """Max-over-ranks wall time and summed counters of one measurement window."""
  if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
    return local_seconds, dict(local_counters)
  import torch
  t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  c = torch.tensor([int(local_counters[k]) for k in names], dtype=torch.int64,
                   device=device)
  dist.all_reduce(c, op=dist.ReduceOp.SUM)
  return float(t.item()), {k: int(v) for k, v in zip(names, c.tolist())}
