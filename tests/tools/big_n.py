"""dev helper: very large batch sanity — N worlds stepped and rendered, sampled worlds vs the oracle."""
import os, sys, time
_TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(_TESTS))
sys.path.insert(0, _TESTS)
import numpy as np, torch, util
from meltingpot_amd import engine as E
name, n = sys.argv[1], int(sys.argv[2])
pack = E.load_pack(name)
eng = E.Engine(pack, n, device=0)
eng.reset()
gen = torch.Generator(device=eng.device); gen.manual_seed(3)
steps = 12
acts = torch.randint(0, eng.num_actions, (steps, n, eng.P), generator=gen, device=eng.device, dtype=torch.int32)
wobs = eng.empty(E.OBS_WORLD_RGB)
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(steps):
  eng.step(acts[s]); eng.observe(E.OBS_WORLD_RGB, wobs)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
rng = np.random.default_rng(0)
for w in sorted({0, n - 1, *map(int, rng.integers(0, n, 30))}):
  o = util.make_oracles(pack, 1, offset=w)[0]; o.reset()
  for s in range(steps):
    o.step(acts[s, w].cpu().numpy())
  assert np.array_equal(wobs[w].cpu().numpy(), o.render_world()), w
print(f"{name} N={n}: {n * eng.P * steps / dt / 1e6:.1f} M agent-steps/s, 32 sampled worlds bit-exact, "
      f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB torch + engine state {n * eng.info.world_state_bytes / 2**30:.2f} GiB")
