"""dev helper: µs per step as the episode goes on (blocks of 50 steps): does a
launch slow down when the map fills with claimed / dirty / eaten cells?"""
import sys
import torch
from meltingpot_amd import engine as E

sub, n, obs_kind = sys.argv[1], int(sys.argv[2]), sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
dev = E.MpDevOptions() if hasattr(E, "MpDevOptions") else None
eng = E.Engine(E.load_pack(sub), n, device=0, auto_reset=True, dev={"verbose": 1})
kind = E.OBS_WORLD_RGB if obs_kind == "world" else E.OBS_RGB
if obs_kind != "none": eng.bind(kind)
gen = torch.Generator(device=eng.device); gen.manual_seed(5)
acts = torch.randint(0, eng.num_actions, (64, n, eng.P), generator=gen, device=eng.device, dtype=torch.int32)
eng.reset()
out = []
for blk in range(steps // 50):
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for i in range(50): eng.step(acts[(blk * 50 + i) % 64])
  b.record(); torch.cuda.synchronize()
  out.append(a.elapsed_time(b) / 50 * 1e3)
print(sub, obs_kind, "us/step per block of 50:", " ".join(f"{t:.0f}" for t in out), flush=True)
