"""Device memory the frame launch writes at full speed, for tensors the CALLER owns.

How fast the engine's one launch per step writes a bound pixel view depends on where the
view's physical pages lie (profiles/r05_alloc_method.md): a view mapped from separately
created 2 MB chunks — what `Engine.bind(kind)` / `Substrate(..., rollout_length=T)` allocate
themselves — was served evenly in 57 of 60 fresh processes over three boxes; an ordinary
`torch.empty` / `hipMalloc` of the same size, physically contiguous in large pieces, was
10 - 15 % slower in about half of them, and a fully contiguous extent 25 - 45 % slower in all.
The reference hands back host arrays (dmlab2d), so it has no counterpart of this module.

A learner that wants to own its rollout buffers and still bind them gets the engine's kind
of memory through torch's own allocator interface:

    from meltingpot_amd import memory
    with memory.mapped_allocations():
      rollout = torch.empty((T, N, P, 88, 88, 3), dtype=torch.uint8, device="cuda")
    engine.bind_ring(OBS_RGB, rollout)

Allocations of 32 MB and more made inside the context come from scattered 2 MB chunks
(libmp_engine.so: mp_torch_alloc / mp_torch_free behind a
`torch.cuda.memory.CUDAPluggableAllocator` in its own `torch.cuda.MemPool`); everything else
about them is torch's (caching, streams, `del`).  Nothing here computes.
"""

from __future__ import annotations

import contextlib

from meltingpot_amd import _build, engine as engine_lib

_pool = None
_allocator = None   # (kept alive: the pool holds a raw pointer into it)


def mapped_pool():
  """The process-wide `torch.cuda.MemPool` whose memory is mapped from scattered 2 MB chunks."""
  global _pool, _allocator
  if _pool is None:
    import torch
    engine_lib.load_library()          # built, and loaded after torch's HIP runtime
    _allocator = torch.cuda.memory.CUDAPluggableAllocator(
        _build.LIB_PATH, "mp_torch_alloc", "mp_torch_free")
    _pool = torch.cuda.MemPool(_allocator.allocator())
  return _pool


@contextlib.contextmanager
def mapped_allocations(device=None):
  """Tensors allocated on `device` inside this context come from `mapped_pool()`."""
  import torch
  with torch.cuda.use_mem_pool(mapped_pool(), device=device):
    yield
