"""dev helper (GPU box): the HOST's cost of one batched `Substrate.step` — device actions, 64
worlds (the launch is far shorter than the Python around it), cProfile of 2000 steps."""
import cProfile, pstats, sys, time
import torch
from meltingpot_amd import substrate
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
env = substrate.build("clean_up", roles=("default",) * 7, num_worlds=n)
eng = env.engine
acts = torch.randint(0, eng.num_actions, (64, n, eng.P), device=eng.device, dtype=torch.int32)
env.reset()
for i in range(200): env.step(acts[i % 64])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2000): env.step(acts[i % 64])
torch.cuda.synchronize()
print(f"{n} worlds: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per Substrate.step (host-bound)")
pr = cProfile.Profile(); pr.enable()
for i in range(2000): env.step(acts[i % 64])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
