"""dev helper (GPU box): find worlds of externality_mushrooms__dense in which a sanctions
marking ends up on the map AWAY from its living avatar (avatar_library.lua:1099-1110 — a
respawn onto another avatar's orphaned marking, or the level reset of a marking that never
came back: the cases round 5 counted instead of restating).

  python tools/gpu_find_displaced_markings.py [worlds] [steps] [first_world]

Actions are a pure function of (global world, step, player) — tests/util.py:hashed_actions —
so a world that is found can
be replayed alone, on the oracle and on an engine created with world_offset = that world
(tests/test_gpu_mushroom.py::test_markings_connected_at_a_distance holds the ones found).
Prints, per hit: world, the 50-step window in which MP_CTR_AUX0 of that world first rose."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import hashed_actions as actions_for   # noqa: E402  (the tests replay with the same function)


def main():
  import torch
  from meltingpot_amd import engine as E, sharding
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
  first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
  pack = E.load_pack("externality_mushrooms__dense")
  eng = E.Engine(pack, n, world_offset=first, auto_reset=False)
  P = eng.P
  stride = eng.info.world_state_bytes
  eng.reset()
  snap = eng.snapshot().reshape(n, stride)
  seed0 = np.array([sharding.world_seed(first)], np.uint64).view(np.uint8)
  hits = [i for i in range(0, stride - 8, 8) if np.array_equal(snap[0, i:i + 8], seed0)]
  assert len(hits) == 1, hits
  ctr5 = hits[0] + 8 + 5 * 4       # WorldTail: seed, then ctr[8]
  worlds = np.arange(first, first + n)
  seen = np.zeros(n, np.uint32)
  found = []
  for s0 in range(0, steps, 50):
    for s in range(s0, min(steps, s0 + 50)):
      eng.step(torch.from_numpy(actions_for(worlds, s, P)).to(eng.device))
    snap = eng.snapshot().reshape(n, stride)
    now = snap[:, ctr5:ctr5 + 4].copy().view(np.uint32)[:, 0]
    for w in np.nonzero((now > 0) & (seen == 0))[0]:
      found.append((int(first + w), s0, int(now[w])))
      print(f"world {first + w}: first away in steps [{s0}, {s0 + 50}), count {now[w]} at the window's end", flush=True)
    seen = np.maximum(seen, now)
    done = snap[:, hits[0] - 320 + 296:hits[0] - 320 + 300].copy().view(np.int32)[:, 0]   # WorldTail::done
    if done.all():
      break
  c = eng.counters()
  print(f"{n} worlds x {s + 1} steps: {len(found)} worlds with a marking away from its avatar; counters {c}")
  eng.close()


if __name__ == "__main__":
  main()
