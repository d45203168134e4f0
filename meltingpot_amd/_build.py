"""Builds libmp_engine.so (hand-written HIP for gfx950) in-tree with hipcc.

The shared library is the product's only compute path; there is no fallback.
`hipcc` cross-compiles without a GPU, so this runs in the CPU container too.
"""

from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmp_engine.so")
SOURCES = ("mp_engine.hip", "step_clean_up.hip", "step_commons.hip", "step_territory.hip", "render.hip")
HEADERS = ("mp_common.h", "step_common.h", "../../include/mp_engine.h", "../../include/mp_pack.h")
ARCH = "gfx950"


def _stale() -> bool:
  if not os.path.exists(LIB_PATH):
    return True
  built = os.path.getmtime(LIB_PATH)
  for f in SOURCES + HEADERS:
    if os.path.getmtime(os.path.join(CSRC, f)) > built:
      return True
  return False


def build_engine(force: bool = False, verbose: bool = False) -> str:
  """Compiles the engine if missing or stale; returns the library path."""
  if not force and not _stale():
    return LIB_PATH
  hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  if not os.path.exists(hipcc):
    raise RuntimeError("hipcc not found: cannot build libmp_engine.so")
  os.makedirs(LIB_DIR, exist_ok=True)
  cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared",
         "-fPIC", "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
  if verbose:
    print(" ".join(cmd))
  subprocess.run(cmd, check=True)
  return LIB_PATH
