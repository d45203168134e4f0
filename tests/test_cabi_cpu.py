"""CPU-side checks of the drop-in boundary: libmp_engine.so builds (hipcc
cross-compiles gfx950 without a GPU), loads, exports every symbol that
include/mp_engine.h declares, and fails loudly — no CPU fallback — when asked
to compute without a device."""
import ctypes

import numpy as np
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
  assert L.mp_abi_version() == engine.MP_ABI_VERSION == 8


def test_library_exports_exactly_the_declared_symbols():
  """A C-ABI boundary exports its ABI and nothing else: the library is compiled with
  -fvisibility=hidden and linked against csrc/exports.map, so `nm -D` shows the header's
  mp_* entry points only (no launch_frame, plan_frame, timed_launches_us ...)."""
  path = _build.build_engine()
  out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True,
                       check=True).stdout
  exported = sorted(line.split()[-1] for line in out.splitlines()
                    if line.split() and line.split()[-2] in ("T", "D", "B", "R"))
  assert exported == _declared_symbols()


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


def _create_rc(blob, num_players=0):
  L = engine.load_library()
  cfg = engine.MpConfig(ctypes.sizeof(engine.MpConfig), 0, 2, 1, 0, 0, None, num_players, 0, 0, 0)
  h = ctypes.c_void_p()
  buf = ctypes.create_string_buffer(bytes(blob), len(blob))
  rc = L.mp_create(buf, len(blob), ctypes.byref(cfg), ctypes.byref(h))
  return rc, L.mp_last_error().decode()


def test_malformed_packs_are_refused_before_a_device_is_touched(clean_up_pack, territory_pack):
  """mp_create takes arbitrary bytes: truncated blobs, overflowing counts, missing
  or mistyped tables and out-of-range indices must come back as MP_ERR_PACK (-2),
  never as a crash.  (Without a GPU a well-formed pack gets as far as
  MP_ERR_NO_DEVICE, -3: everything below is checked on the host first.)"""
  import struct
  import torch
  from meltingpot_amd import lower, pack
  good = -3 if not torch.cuda.is_available() else 0
  rc, _ = _create_rc(clean_up_pack)
  assert rc == good
  # truncation, bad magic
  assert _create_rc(clean_up_pack[:-16])[0] == -2
  assert _create_rc(b"XPK1" + clean_up_pack[4:])[0] == -2
  # an entry whose count * size overflows 64 bits / whose offset leaves the blob
  hdr_size, entry_size = 16, 64
  blob = bytearray(clean_up_pack)
  struct.pack_into("<Q", blob, hdr_size + 40, 0xFFFFFFFFFFFFFFF0)   # entry 0: count
  assert _create_rc(blob)[0] == -2
  blob = bytearray(clean_up_pack)
  struct.pack_into("<Q", blob, hdr_size + 48, len(blob) + 16)       # entry 0: offset
  assert _create_rc(blob)[0] == -2
  blob = bytearray(clean_up_pack)
  struct.pack_into("<Q", blob, hdr_size + 48, 8)                    # inside the entry table
  assert _create_rc(blob)[0] == -2
  # missing / mistyped / short tables, indices out of range
  t = pack.loads(clean_up_pack)
  for name in ("state_layer", "apple_cells", "view_sprite_map", "init_spawn_mask", "cu_states"):
    bad = {k: v for k, v in t.items() if k != name}
    rc, msg = _create_rc(pack.dumps(bad))
    assert rc == -2, (name, rc, msg)
  bad = dict(t); bad["state_layer"] = t["state_layer"].astype(np.float64)
  assert _create_rc(pack.dumps(bad))[0] == -2
  bad = dict(t); bad["init_grid"] = t["init_grid"][:-5]
  assert _create_rc(pack.dumps(bad))[0] == -2
  bad = dict(t); bad["apple_cells"] = t["apple_cells"].copy(); bad["apple_cells"][3] = 21 * 30
  assert _create_rc(pack.dumps(bad))[0] == -2
  bad = dict(t); bad["spawn_cells"] = t["spawn_cells"].copy(); bad["spawn_cells"][0] = -1
  assert _create_rc(pack.dumps(bad))[0] == -2
  bad = dict(t); bad["init_grid"] = t["init_grid"].copy(); bad["init_grid"][7] = 250
  assert _create_rc(pack.dumps(bad))[0] == -2
  bad = dict(t); bad["cu_i32"] = t["cu_i32"].copy(); bad["cu_i32"][5] = 0     # ee_interval: % 0
  rc, msg = _create_rc(pack.dumps(bad))
  assert rc in (-2, good)   # a rule constant: checked with the device's tables when there is one
  bad = dict(t); bad["hdr"] = t["hdr"].copy(); bad["hdr"][lower.HDR_NHITS] = 30
  assert _create_rc(pack.dumps(bad))[0] == -2
  # player counts
  assert _create_rc(clean_up_pack, num_players=16)[0] == -1
  assert _create_rc(clean_up_pack, num_players=15)[0] == good
  t3 = pack.loads(territory_pack)
  bad = dict(t3); bad["resource_cells"] = t3["resource_cells"].copy(); bad["resource_cells"][0] = 10**6
  rc, msg = _create_rc(pack.dumps(bad))
  assert rc in (-2, good)


def test_malformed_matrix_packs_are_refused():
  """The *_in_the_matrix tables: lengths follow from R and the colour intervals,
  classes and states are range-checked, all on the host."""
  import torch
  from meltingpot_amd import engine, pack
  good = -3 if not torch.cuda.is_available() else 0
  blob = engine.load_pack("running_with_scissors_in_the_matrix__arena")
  assert _create_rc(blob)[0] == good
  t = pack.loads(blob)
  for name in ("mx_i32", "mx_f64", "mx_states", "mx_thr", "mx_player_i32", "mx_player_f64",
               "resource_class", "resource_cells"):
    bad = {k: v for k, v in t.items() if k != name}
    rc, msg = _create_rc(pack.dumps(bad))
    assert rc == -2, (name, rc, msg)
  def variant(**kw):
    bad = dict(t)
    for k, fn in kw.items():
      v = t[k].copy(); fn(v); bad[k] = v
    return _create_rc(pack.dumps(bad))[0]
  assert variant(mx_i32=lambda v: v.__setitem__(0, 4)) == -2       # R > 3
  assert variant(mx_i32=lambda v: v.__setitem__(0, 2)) == -2       # R that the tables do not fit
  assert variant(mx_i32=lambda v: v.__setitem__(16, 0)) == -2      # intervalLength 0: % 0
  assert variant(mx_i32=lambda v: v.__setitem__(18, 7)) == -2      # initialHealth > 3 (2 bits)
  assert variant(mx_i32=lambda v: v.__setitem__(21, 9)) == -2      # hit index
  assert variant(resource_class=lambda v: v.__setitem__(5, 4)) == -2
  assert variant(resource_cells=lambda v: v.__setitem__(0, 24 * 25)) == -2
  assert variant(mx_states=lambda v: v.__setitem__(3, 255)) == -2
  bad = dict(t); bad["mx_f64"] = t["mx_f64"][:-1]
  assert _create_rc(pack.dumps(bad))[0] == -2
  bad = dict(t); bad["resource_class"] = t["resource_class"][:-1]
  assert _create_rc(pack.dumps(bad))[0] == -2


def test_malformed_mushroom_packs_are_refused():
  """externality_mushrooms: tables missing, short or leaving the map are refused on the host;
  the rule constants with the device's tables (tests/test_gpu_mushroom.py)."""
  import torch
  from meltingpot_amd import engine, pack
  good = -3 if not torch.cuda.is_available() else 0
  blob = engine.load_pack("externality_mushrooms__dense")
  assert _create_rc(blob)[0] == good
  t = pack.loads(blob)
  for name in ("em_states", "em_i32", "em_f64", "em_thr", "mushroom_cells", "zapper_i32"):
    bad = {k: v for k, v in t.items() if k != name}
    rc, msg = _create_rc(pack.dumps(bad))
    assert rc == -2, (name, rc, msg)
  for name, keep in (("em_i32", 29), ("em_thr", 20), ("em_states", 7)):
    bad = dict(t); bad[name] = t[name][:keep]
    assert _create_rc(pack.dumps(bad))[0] == -2, name
  bad = dict(t); bad["mushroom_cells"] = t["mushroom_cells"].copy(); bad["mushroom_cells"][7] = 14 * 23
  assert _create_rc(pack.dumps(bad))[0] == -2
  bad = dict(t); bad["mushroom_cells"] = np.concatenate([t["mushroom_cells"]] * 2)   # 462 sites: four a lane
  assert _create_rc(pack.dumps(bad))[0] == -2

