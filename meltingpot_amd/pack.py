"""MPK1: the lowered-substrate table container shared by host, engine and oracle.

A pack is a flat, little-endian, position-independent blob:

    header   : char magic[4] = "MPK1"; u32 n_entries; u64 total_bytes
    entries  : n_entries x { char name[32]; u32 dtype; u32 reserved;
                             u64 count; u64 offset; u64 reserved2 }   (64 B)
    payload  : arrays, each 16-byte aligned, `offset` from blob start

dtype codes: 0=u8 1=i32 2=f64 3=u64 4=u32.  The C-side reader is
`include/mp_pack.h`.  A pack carries the *data* half of the reference's
substrate definition (ASCII map, prefabs, sprites, component kwargs: reference
`meltingpot/configs/substrates/*.py`) after lowering (`lower.py`).
"""

from __future__ import annotations

import struct
from typing import Dict

import numpy as np

MAGIC = b"MPK1"
_DTYPES = {
    np.dtype("uint8"): 0,
    np.dtype("int32"): 1,
    np.dtype("float64"): 2,
    np.dtype("uint64"): 3,
    np.dtype("uint32"): 4,
}
_CODES = {v: k for k, v in _DTYPES.items()}
_ENTRY = struct.Struct("<32sIIQQQ")
_HEADER = struct.Struct("<4sIQ")


def dumps(tables: Dict[str, np.ndarray]) -> bytes:
  names = sorted(tables)
  n = len(names)
  offset = _HEADER.size + n * _ENTRY.size
  offset = (offset + 15) & ~15
  entries = []
  payload = []
  for name in names:
    arr = np.ascontiguousarray(tables[name])
    if arr.dtype not in _DTYPES:
      raise TypeError(f"{name}: unsupported dtype {arr.dtype}")
    raw = arr.tobytes()
    bname = name.encode()
    if len(bname) > 31:
      raise ValueError(f"table name too long: {name}")
    entries.append(_ENTRY.pack(bname, _DTYPES[arr.dtype], 0, arr.size, offset,
                               0))
    pad = (-len(raw)) & 15
    payload.append(raw + b"\0" * pad)
    offset += len(raw) + pad
  head = _HEADER.pack(MAGIC, n, offset)
  blob = head + b"".join(entries)
  blob += b"\0" * ((-len(blob)) & 15)
  blob += b"".join(payload)
  assert len(blob) == offset
  return blob


def loads(blob: bytes) -> Dict[str, np.ndarray]:
  magic, n, total = _HEADER.unpack_from(blob, 0)
  if magic != MAGIC:
    raise ValueError("not an MPK1 pack")
  if total != len(blob):
    raise ValueError(f"pack truncated: header says {total}, got {len(blob)}")
  out = {}
  for i in range(n):
    bname, code, _, count, offset, _ = _ENTRY.unpack_from(
        blob, _HEADER.size + i * _ENTRY.size)
    name = bname.rstrip(b"\0").decode()
    dt = _CODES[code]
    out[name] = np.frombuffer(blob, dtype=dt, count=count,
                              offset=offset).copy()
  return out
