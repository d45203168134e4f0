"""dev helper: is the two-speed behaviour of the per-agent launches a property of
the PROCESS (buffer placement) or of the moment?  One process, several engines and
several placements of the bound view inside one pool; 5 x 60 steps each."""
import sys, time
import torch
from meltingpot_amd import engine as E

sub, n = sys.argv[1], int(sys.argv[2])
pack = E.load_pack(sub)
dev = torch.device("cuda", 0)
pool = None
for trial in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
  eng = E.Engine(pack, n, device=0, auto_reset=True)
  shape, dtype = eng.shapes[E.OBS_RGB]
  nbytes = 1
  for d in shape: nbytes *= d
  if pool is None:
    pool = torch.empty(nbytes + (64 << 20), dtype=torch.uint8, device=dev)
  off = [0, 4096, 65536, 1 << 20, (2 << 20) + 4096 * 3, 256, 0, 0][trial % 8]
  obs = pool[off:off + nbytes].view(shape)
  eng.bind(E.OBS_RGB, obs)
  gen = torch.Generator(device=dev); gen.manual_seed(5)
  acts = torch.randint(0, eng.num_actions, (64, n, eng.P), generator=gen, device=dev, dtype=torch.int32)
  eng.reset()
  for i in range(20): eng.step(acts[i % 64])
  ts = []
  for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for i in range(60): eng.step(acts[i % 64])
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 60 * 1e3)
  print(f"{sub} trial {trial} obs ptr {obs.data_ptr():#x} (off {off}) us/step: " + " ".join(f"{t:.1f}" for t in ts), flush=True)
  eng.close(); del eng
