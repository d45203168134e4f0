"""dev helper: per-step GPU time (one event pair a step) of the first steps after a reset
— what does a short timed region (bench.py --steps 20 --warmup 5) see?"""
import sys, time
import torch
from meltingpot_amd import engine as E
sub, n, view = sys.argv[1], int(sys.argv[2]), sys.argv[3]
kind = E.OBS_WORLD_RGB if view == "world" else E.OBS_RGB
eng = E.Engine(E.load_pack(sub), n, device=0, auto_reset=True)
eng.bind(kind)
print("placement:", eng.placement[kind]["kind"], min(eng.placement[kind]["dry_launch_us"]))
gen = torch.Generator(device=eng.device); gen.manual_seed(1234)
acts = torch.randint(0, eng.num_actions, (256, n, eng.P), generator=gen, device=eng.device, dtype=torch.int32)
for trial, idle in enumerate((0.0, 0.0, 1.0)):
  eng.reset()
  torch.cuda.synchronize()
  time.sleep(idle)
  K = 400
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
  ev[0].record()
  for i in range(K):
    eng.step(acts[i % 256]); ev[i + 1].record()
  torch.cuda.synchronize()
  ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
  blocks = [sum(ts[i:i + 10]) / 10 for i in range(0, 100, 10)] + [sum(ts[i:i + 50]) / 50 for i in range(100, K, 50)]
  print(f"trial {trial} (idle {idle} s before): steps 0-99 by tens:", " ".join(f"{b:.0f}" for b in blocks[:10]),
        "| then by fifties:", " ".join(f"{b:.0f}" for b in blocks[10:]), flush=True)
