import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meltingpot_amd import engine as E
for sub, n in (("clean_up", 4096), ("commons_harvest__open", 4096), ("territory__rooms", 8192)):
  eng = E.Engine(E.load_pack(sub), n)
  eng.reset()
  a = torch.randint(0, eng.num_actions, (8, n, eng.P), device=eng.device, dtype=torch.int32)
  for i in range(3): eng.step(a[i])
  eng.sync(); print("====", sub, flush=True)
  eng.close()
