"""dev helper: paired A/B of engine builds INSIDE one process — the same eight
buffers for the bound view, every build timed on every buffer in turn, so the
placement of the output (profiles/r03_buffer_placement.md: +-12 %) cancels out.

  python tools/gpu_paired_ab.py <substrate> <worlds> <world|agents> <libA> <libB> ...
  (lib = a tag of meltingpot_amd/lib/libmp_engine_<tag>.so, "-" = the product build;
  "<lib>:max_groups=192,feeders=3" adds MpDevOptions for that engine)"""
import os, sys
import torch
from meltingpot_amd import engine as E

sub, n, view = sys.argv[1], int(sys.argv[2]), sys.argv[3]
tags = sys.argv[4:]
root = os.path.dirname(os.path.abspath(E.__file__))
kind = E.OBS_WORLD_RGB if view == "world" else E.OBS_RGB
both = view == "both"    # per-agent RGB on the varied buffers + WORLD.RGB on a fixed one: one launch draws both
pack = E.load_pack(sub)
warm = int(os.environ.get("WARM", "10"))
engines = []
for spec in tags:
  tag, _, plan = spec.partition(":")      # "<lib>[:k=v,k=v]" — MpDevOptions of that engine
  dev = {k: int(v) for k, v in (kv.split("=") for kv in plan.split(",") if kv)}
  E._lib = None
  if tag == "-":
    os.environ.pop("MP_ENGINE_LIB", None)
  else:
    os.environ["MP_ENGINE_LIB"] = os.path.join(root, "lib", f"libmp_engine_{tag}.so")
  eng = E.Engine(pack, n, device=0, auto_reset=True, dev=dev or None)
  eng.reset()
  engines.append(eng)
# NBUF torch allocations + MAPPED views mapped from 2 MB physical chunks (another
# scatter of the same bytes: profiles/r03_buffer_placement.md)
bufs = [engines[0].empty(kind) for _ in range(int(os.environ.get("NBUF", "8")))]
for _ in range(int(os.environ.get("MAPPED", "0"))):
  b = engines[0].empty_mapped(kind, 2 << 20)
  if b is not None:
    bufs.append(b)
# CONTIG physically contiguous extents (hipExtMallocWithFlags(hipDeviceMallocContiguous)): the
# deterministic worst case of profiles/r05_alloc_method.md, 43 - 60 % slower under the stock plan
if int(os.environ.get("CONTIG", "0")):
  import ctypes
  hip = ctypes.CDLL("libamdhip64.so")
  shape, dtype = engines[0].shapes[kind]
  nbytes = 1
  for d in shape: nbytes *= int(d)
  for _ in range(int(os.environ["CONTIG"])):
    ptr = ctypes.c_void_p()
    if hip.hipExtMallocWithFlags(ctypes.byref(ptr), ctypes.c_size_t(nbytes), ctypes.c_uint(0x4)) != 0 or not ptr.value:
      print("(no physically contiguous extent)"); break
    class _Owner:
      __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr.value, False), "version": 2}
    keep = _Owner()
    bufs.append(torch.as_tensor(keep, device=engines[0].device).view(dtype).view(tuple(shape)))
print("buffers:", os.environ.get("NBUF", "8"), "torch +", os.environ.get("MAPPED", "0"), "mapped 2 MB +",
      os.environ.get("CONTIG", "0"), "contiguous extent(s)")
gen = torch.Generator(device=engines[0].device); gen.manual_seed(5)
acts = torch.randint(0, engines[0].num_actions, (64, n, engines[0].P), generator=gen,
                     device=engines[0].device, dtype=torch.int32)
other = engines[0].empty(E.OBS_WORLD_RGB) if both else None
for eng in engines:           # the same episode progress for all
  if both:
    eng.bind(E.OBS_WORLD_RGB, other)
  eng.bind(kind, bufs[0])
  for i in range(warm): eng.step(acts[i % 64])
rows = []
for buf in bufs:
  row = []
  for eng in engines:
    eng.bind(kind, buf)
    for i in range(6): eng.step(acts[i])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(60): eng.step(acts[i % 64])
    b.record(); torch.cuda.synchronize()
    row.append(a.elapsed_time(b) / 60 * 1e3)
  rows.append(row)
print(f"{sub} {view} x{n}: us per step by buffer, builds {tags}")
for row in rows: print("   " + "  ".join(f"{t:6.1f}" for t in row))
print("   min  " + "  ".join(f"{min(r[j] for r in rows):6.1f}" for j in range(len(tags))))
print("   max  " + "  ".join(f"{max(r[j] for r in rows):6.1f}" for j in range(len(tags))))
means = [sum(r[j] for r in rows) / len(rows) for j in range(len(tags))]
print("   mean " + "  ".join(f"{m:6.1f}" for m in means) + "   vs first: " +
      "  ".join(f"{m / means[0]:.3f}" for m in means), flush=True)
