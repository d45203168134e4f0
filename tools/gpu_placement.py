"""dev helper: the first engine of a process steps slower than the second — time
(power state), allocation order, or something the first engine does?"""
import os, sys, time
import torch
from meltingpot_amd import engine as E

sub, n, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
KIND = E.OBS_WORLD_RGB if os.environ.get("VIEW") == "world" else E.OBS_RGB
pack = E.load_pack(sub)
def rounds(eng, tag, k=4):
  gen = torch.Generator(device=eng.device); gen.manual_seed(5)
  acts = torch.randint(0, eng.num_actions, (64, n, eng.P), generator=gen, device=eng.device, dtype=torch.int32)
  for r in range(k):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    evs[0].record()
    for i in range(60):
      eng.step(acts[i % 64]); evs[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(60))
    print(f"[{mode}] {tag} round {r}: min {ts[0]:.0f} p25 {ts[15]:.0f} median {ts[30]:.0f} p75 {ts[45]:.0f} max {ts[-1]:.0f}", flush=True)
import os
DEV = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("DEVPLAN", "").split(",") if kv)}
def make():
  eng = E.Engine(pack, n, device=0, auto_reset=True, dev=DEV or None)
  eng.bind(KIND); eng.reset()
  return eng
if mode == "same_engine_later":
  eng = make(); rounds(eng, "first"); time.sleep(2.0); rounds(eng, "after 2 s idle")
  junk = torch.empty(1 << 28, device="cuda")
  t0 = time.time()
  while time.time() - t0 < 1.5: junk.mul_(1.0001)
  torch.cuda.synchronize(); rounds(eng, "after 1.5 s of busy GPU")
elif mode == "create_twice":
  eng = make(); eng.close(); del eng; torch.cuda.empty_cache()
  eng = make(); rounds(eng, "second engine, first never stepped")
elif mode == "busy_first":
  junk = torch.empty(1 << 28, device="cuda")
  t0 = time.time()
  while time.time() - t0 < 1.5: junk.mul_(1.0001)
  torch.cuda.synchronize()
  eng = make(); rounds(eng, "first engine after 1.5 s of busy GPU")
elif mode == "two_alive":
  a = make(); b = make()
  rounds(a, "engine a (b alive)"); rounds(b, "engine b (a alive)"); rounds(a, "engine a again")
elif mode == "rebind":
  eng = make(); obs1 = eng._bound[KIND]
  rounds(eng, f"obs1 {obs1.data_ptr():#x}", 2)
  obs2 = torch.empty_like(obs1); eng.bind(KIND, obs2)
  rounds(eng, f"obs2 {obs2.data_ptr():#x}", 2)
  eng.bind(KIND, obs1); rounds(eng, "obs1 again", 2)
  del obs2
elif mode == "new_engine_old_obs":
  eng = make(); obs1 = eng._bound[KIND]
  rounds(eng, "engine 0", 2)
  eng.close(); del eng
  eng = E.Engine(pack, n, device=0, auto_reset=True); eng.bind(KIND, obs1); eng.reset()
  rounds(eng, "engine 1, engine 0's tensor", 2)
elif mode == "like_bimodal2":
  for inst in range(3):
    eng = make(); obs = eng._bound[KIND]
    rounds(eng, f"engine {inst} obs {obs.data_ptr():#x}", 3)
    eng.close(); del eng, obs
    torch.cuda.empty_cache()
elif mode == "like_bimodal2_no_empty_cache":
  for inst in range(3):
    eng = make(); obs = eng._bound[KIND]
    rounds(eng, f"engine {inst} obs {obs.data_ptr():#x}", 3)
    eng.close(); del eng, obs
elif mode == "many_buffers":
  eng = make(); obs0 = eng._bound[KIND]
  bufs = [obs0] + [torch.empty_like(obs0) for _ in range(7)]
  gen = torch.Generator(device=eng.device); gen.manual_seed(5)
  acts = torch.randint(0, eng.num_actions, (64, n, eng.P), generator=gen, device=eng.device, dtype=torch.int32)
  def timed(f, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps): f(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
  for rep in range(1):
    for j, buf in enumerate(bufs):
      eng.bind(KIND, buf)
      for i in range(10): eng.step(acts[i])
      ts = timed(lambda i: eng.step(acts[i % 64]), 60)
      print(f"[many] buffer {j} {buf.data_ptr():#x}: step {ts:.0f} us", flush=True)
elif mode == "strides":
  big = 4608
  proto = E.Engine(pack, big, device=0, auto_reset=True)
  shape, dtype = proto.shapes[KIND]
  proto.close(); del proto
  per_world = 1
  for d in shape[1:]: per_world *= d
  bufs = [torch.empty(big * per_world, dtype=torch.uint8, device="cuda") for _ in range(6)]
  for nn in (4096, 3840, 4608, 4096):
    eng = E.Engine(pack, nn, device=0, auto_reset=True); eng.reset()
    gen = torch.Generator(device=eng.device); gen.manual_seed(5)
    acts = torch.randint(0, eng.num_actions, (64, nn, eng.P), generator=gen, device=eng.device, dtype=torch.int32)
    out = []
    for j, buf in enumerate(bufs):
      eng.bind(KIND, buf[:nn * per_world].view((nn,) + tuple(shape[1:])))
      for i in range(10): eng.step(acts[i])
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      torch.cuda.synchronize(); a.record()
      for i in range(60): eng.step(acts[i % 64])
      b.record(); torch.cuda.synchronize()
      out.append(a.elapsed_time(b) / 60 * 1e3)
    print(f"[strides] N {nn}: us/step per buffer " + " ".join(f"{t:.0f}" for t in out) +
          "   ns per world " + " ".join(f"{t * 1e3 / nn:.1f}" for t in out), flush=True)
    eng.close(); del eng
elif mode == "prof_buffers":
  eng = make(); obs0 = eng._bound[KIND]
  bufs = [obs0] + [torch.empty_like(obs0) for _ in range(7)]
  gen = torch.Generator(device=eng.device); gen.manual_seed(5)
  acts = torch.randint(0, eng.num_actions, (64, n, eng.P), generator=gen, device=eng.device, dtype=torch.int32)
  for j, buf in enumerate(bufs):
    eng.bind(KIND, buf)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(20): eng.step(acts[i % 64])
    b.record(); torch.cuda.synchronize()
    print(f"[prof] buffer {j} {buf.data_ptr():#x}: {a.elapsed_time(b) / 20 * 1e3:.0f} us/step (20 steps)", flush=True)
elif mode == "probe_corr":
  # does the draw-only launch (mp_observe: no state change) rank the buffers like
  # the fused step launch does?
  eng = make(); obs0 = eng._bound[KIND]
  bufs = [obs0] + [torch.empty_like(obs0) for _ in range(7)]
  gen = torch.Generator(device=eng.device); gen.manual_seed(5)
  acts = torch.randint(0, eng.num_actions, (64, n, eng.P), generator=gen, device=eng.device, dtype=torch.int32)
  def timed(f, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(reps): f(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
  for i in range(30): eng.step(acts[i])
  for j, buf in enumerate(bufs):
    eng.unbind(KIND)
    for i in range(3): eng.observe(KIND, buf)
    t_draw = timed(lambda i: eng.observe(KIND, buf), 10)
    flat = buf.view(-1)
    t_fill = timed(lambda i: flat.fill_(i & 255), 5)
    eng.bind(KIND, buf)
    for i in range(10): eng.step(acts[i])
    t_step = timed(lambda i: eng.step(acts[i % 64]), 60)
    print(f"[probe] buffer {j}: draw-only {t_draw:.0f} us, fused step {t_step:.0f} us, fill {t_fill:.0f} us", flush=True)
