"""CPU-side checks of the drop-in boundary: libmp_engine.so builds (hipcc
cross-compiles gfx950 without a GPU), loads, exports every symbol that
include/mp_engine.h declares, and fails loudly — no CPU fallback — when asked
to compute without a device."""
import ctypes
import os
import re
import subprocess

import pytest

from meltingpot_amd import _build, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
  text = open(os.path.join(ROOT, "include", "mp_engine.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(mp_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
  assert _declared_symbols() == sorted(engine.ABI_SYMBOLS)


def test_library_builds_loads_and_exports_every_declared_symbol():
  L = engine.load_library()
  for name in _declared_symbols():
    assert hasattr(L, name), f"{name} declared in mp_engine.h but not exported"
  assert L.mp_abi_version() == engine.MP_ABI_VERSION == 2


def test_library_contains_gfx950_code_object():
  path = _build.build_engine()
  out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", path],
                       capture_output=True, text=True).stdout
  assert "gfx950" in out
  # the three hot kernels are in the fat binary
  blob = open(path, "rb").read()
  for kernel in (b"k_step_clean_up", b"k_step_commons", b"k_step_territory", b"k_frame"):
    assert kernel in blob


def test_no_cpu_fallback(clean_up_pack):
  import torch
  if torch.cuda.is_available():
    pytest.skip("a GPU is present")
  with pytest.raises(engine.EngineError, match="no HIP device"):
    engine.Engine(clean_up_pack, 2)


def test_bad_arguments_are_rejected_before_touching_a_device(clean_up_pack):
  L = engine.load_library()
  cfg = engine.MpConfig(ctypes.sizeof(engine.MpConfig), 0, 0, 1, 0, 0, None)
  h = ctypes.c_void_p()
  buf = ctypes.create_string_buffer(clean_up_pack, len(clean_up_pack))
  assert L.mp_create(buf, len(clean_up_pack), ctypes.byref(cfg), ctypes.byref(h)) == -1
  assert b"num_worlds" in L.mp_last_error()
  cfg.num_worlds = 4
  junk = ctypes.create_string_buffer(b"not a pack" * 10, 100)
  assert L.mp_create(junk, 100, ctypes.byref(cfg), ctypes.byref(h)) == -2
  cfg.struct_size = 3
  assert L.mp_create(buf, len(clean_up_pack), ctypes.byref(cfg), ctypes.byref(h)) == -1


def test_product_never_imports_the_oracle():
  """oracle/ is test infrastructure: nothing under meltingpot_amd/ may touch it."""
  pkg = os.path.join(ROOT, "meltingpot_amd")
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".h")):
        src = open(os.path.join(dirpath, f)).read()
        assert "import oracle" not in src and "from oracle" not in src, f
        assert "liboracle" not in src, f
