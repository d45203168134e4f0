"""dev helper: where do the two speeds of a per-agent launch come from?  One engine,
individual step times (one event pair per step) in rounds separated by idle gaps;
then a second engine in the same process."""
import sys, time
import torch
from meltingpot_amd import engine as E

sub, n = sys.argv[1], int(sys.argv[2])
pack = E.load_pack(sub)
def rounds(eng, tag, k=8, gap=0.3):
  gen = torch.Generator(device=eng.device); gen.manual_seed(5)
  acts = torch.randint(0, eng.num_actions, (64, n, eng.P), generator=gen, device=eng.device, dtype=torch.int32)
  for r in range(k):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    evs[0].record()
    for i in range(60):
      eng.step(acts[i % 64]); evs[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(60))
    print(f"{tag} round {r}: min {ts[0]:.0f} p25 {ts[15]:.0f} median {ts[30]:.0f} p75 {ts[45]:.0f} max {ts[-1]:.0f}", flush=True)
    time.sleep(gap if r % 2 == 0 else 0.0)
for inst in range(2):
  eng = E.Engine(pack, n, device=0, auto_reset=True)
  obs = eng.bind(E.OBS_RGB)
  eng.reset()
  rounds(eng, f"{sub} engine {inst} obs {obs.data_ptr():#x}")
  eng.close(); del eng, obs
  torch.cuda.empty_cache()
