"""dev helper: per-step latency of the unbatched drop-in API (one world, numpy leaves)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from meltingpot_amd import substrate

name = sys.argv[1] if len(sys.argv) > 1 else "clean_up"
cfg = substrate.get_config(name)
with substrate.build(name, roles=cfg.default_player_roles) as env:
  ts = env.reset()
  n = len(cfg.default_player_roles)
  rng = np.random.default_rng(0)
  acts = rng.integers(0, len(cfg.action_set), size=(300, n))
  for i in range(50):
    ts = env.step(acts[i])
  t0 = time.perf_counter()
  for i in range(50, 300):
    ts = env.step(acts[i])
  dt = (time.perf_counter() - t0) / 250
  print(f"{name}: {dt * 1e6:.0f} us per env.step ({1 / dt:.0f} steps/s, "
        f"{n / dt:.0f} agent-steps/s), leaves: {type(ts.observation[0]['RGB']).__name__}")
