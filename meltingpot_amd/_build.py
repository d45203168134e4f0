"""Builds libmp_engine.so (hand-written HIP for gfx950) in-tree with hipcc.

The shared library is the product's only compute path; there is no fallback.
`hipcc` cross-compiles without a GPU, so this runs in the CPU container too.
"""

from __future__ import annotations

import fcntl
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmp_engine.so")
SOURCES = ("mp_engine.hip", "step_kernels.hip", "frame.hip")
HEADERS = ("mp_common.h", "step_common.h", "step_clean_up.h", "step_commons.h",
           "step_territory.h", "step_coins.h", "step_matrix.h", "step_coop.h", "step_gift.h", "step_cook.h", "step_mushroom.h", "../../include/mp_engine.h",
           "../../include/mp_pack.h", "exports.map")
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
  # One builder at a time (N ranks of `bench.py --gpus N` import this together),
  # and readers never see a half-written library: compile aside, then rename.
  with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
    fcntl.flock(lock, fcntl.LOCK_EX)
    if not force and not _stale():  # another process built it while we waited
      return LIB_PATH
    tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
    # one object per source (kept under lib/obj: a change to mp_engine.hip does not
    # recompile the frame kernel's 40 s), then one link
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    objects, jobs = [], []
    for src in SOURCES:
      path = os.path.join(CSRC, src)
      obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
      objects.append(obj)
      if (force or not os.path.exists(obj) or
          os.path.getmtime(obj) < max(os.path.getmtime(path), newest_header)):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
               "-fvisibility=hidden", "-c", "-o", obj + f".{os.getpid()}.tmp", path]
        if verbose:
          print(" ".join(cmd))
        jobs.append((obj, subprocess.Popen(cmd)))
    failed = [obj for obj, pr in jobs if pr.wait() != 0]
    for obj, _ in jobs:
      part = obj + f".{os.getpid()}.tmp"
      if obj not in failed and os.path.exists(part):
        os.replace(part, obj)
      elif os.path.exists(part):
        os.remove(part)
    if failed:
      raise RuntimeError(f"hipcc failed on {[os.path.basename(o) for o in failed]}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC",
           f"-Wl,--version-script={os.path.join(CSRC, 'exports.map')}", "-o", tmp] + objects
    if verbose:
      print(" ".join(cmd).replace(tmp, LIB_PATH))
    try:
      subprocess.run(cmd, check=True)
      os.replace(tmp, LIB_PATH)
    finally:
      # (this hipcc leaves its offload-bundle intermediates next to the output)
      for name in os.listdir(LIB_DIR):
        if name.startswith(os.path.basename(tmp)) or ".hipv4-" in name or ".host-x86_64-" in name:
          os.remove(os.path.join(LIB_DIR, name))
  return LIB_PATH
