"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

GOLDEN = 0x9E3779B97F4A7C15
MASK64 = (1 << 64) - 1


def world_seed(w: int) -> int:
  """Default per-world seed of the engine."""
  return (GOLDEN * (w + 1)) & MASK64


def make_oracles(pack_bytes, n, offset=0):
  from oracle import oracle
  return [oracle.Oracle(pack_bytes, world_seed(offset + w)) for w in range(n)]


def random_actions(rng, steps, n, p, nact, weights=None):
  if weights is None:
    return rng.integers(0, nact, size=(steps, n, p), dtype=np.int32)
  w = np.asarray(weights, np.float64)
  return rng.choice(nact, size=(steps, n, p), p=w / w.sum()).astype(np.int32)


def patch_pack(pack_bytes, **hdr_overrides):
  """Returns a pack with some header fields replaced (e.g. MAXFRAMES)."""
  from meltingpot_amd import lower, pack
  t = pack.loads(pack_bytes)
  for k, v in hdr_overrides.items():
    t["hdr"][getattr(lower, "HDR_" + k)] = v
  return pack.dumps(t)
