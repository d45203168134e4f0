"""dev helper (GPU box): one world of a big batch against the oracle after EVERY step.

  python tools/gpu_trace_world.py <pack> <world> [worlds] [steps] [view: world|agents|none] [alone]

Runs the batch the way tests/tools/deep_soak.py does (hashed actions, no auto-reset, a
view bound, the fused launch) and holds world `world` — its state record, rewards, events and
the bound view — against an oracle stepped beside it; stops at the first step that differs and
says what differs.  With `alone`, the engine holds only the sixteen worlds around `world`
(world_offset): does the difference need the big batch?"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import numpy as np
import torch
import util
from meltingpot_amd import engine as E
from oracle import oracle


def main():
  sub, w = sys.argv[1], int(sys.argv[2])
  n = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
  steps = int(sys.argv[4]) if len(sys.argv) > 4 else 900
  view = sys.argv[5] if len(sys.argv) > 5 else "agents"
  alone = len(sys.argv) > 6 and sys.argv[6] == "alone"
  pack = E.load_pack(sub)
  first = (w & ~15) if alone else 0
  if alone:
    n = 16
  eng = E.Engine(pack, n, device=0, auto_reset=False, unfused=False, placements=1, world_offset=first)
  kind = {"world": E.OBS_WORLD_RGB, "agents": E.OBS_RGB}.get(view)
  bound = eng.bind(kind) if kind is not None else None
  o = oracle.Oracle(pack, util.world_seed(w), 0)
  o.reset()
  eng.reset()
  worlds = np.arange(first, first + n)
  i = w - first
  print(f"{sub}: world {w} of [{first}, {first + n}), {steps} steps, view {view}, fused {eng.fused}, plan {eng.plan}")
  for s in range(steps):
    acts = util.hashed_actions(worlds, s, eng.P, num_actions=eng.num_actions)
    eng.step(torch.from_numpy(acts).to(eng.device))
    o.step(acts[i])
    grid, avat, glob_ = eng.dump()
    og, oa, ogl = o.dump()
    rew = eng.observe(E.OBS_REWARD).cpu().numpy()[i]
    ev = eng.observe(E.OBS_EVENTS).cpu().numpy()[i]
    got_ev = sorted(tuple(int(v) for v in r[:3]) for r in ev[1:1 + int(ev[0, 0])])
    diffs = []
    if not np.array_equal(grid[i], og):
      where = np.argwhere(grid[i] != og)
      diffs.append(f"grid at {where[:12].tolist()} engine {grid[i][tuple(where[:12].T)].tolist()} "
                   f"oracle {og[tuple(where[:12].T)].tolist()}")
    if not np.array_equal(avat[i], oa):
      diffs.append(f"avatars engine {avat[i].tolist()} oracle {oa.tolist()}")
    if not np.array_equal(glob_[i], ogl):
      diffs.append(f"globals engine {glob_[i].tolist()} oracle {ogl.tolist()}")
    if not np.array_equal(rew, o.rewards()):
      diffs.append(f"rewards engine {rew.tolist()} oracle {o.rewards().tolist()}")
    if got_ev != sorted(o.events()):
      diffs.append(f"events engine {got_ev} oracle {sorted(o.events())}")
    if bound is not None:
      ov = o.render_world() if view == "world" else np.stack([o.render_agent(p) for p in range(o.P)])
      gv = bound[i].cpu().numpy()
      if not np.array_equal(gv, ov):
        where = np.argwhere(gv != ov)
        diffs.append(f"view: {len(where)} bytes differ, first at {where[0].tolist()}, last at {where[-1].tolist()}")
    if diffs:
      print(f"step {s} (actions {acts[i].tolist()}): DIFFERS")
      for d in diffs:
        print("  " + d)
      print(f"  engine globals {glob_[i].tolist()}\n  oracle globals {ogl.tolist()}")
      print(f"  engine avatars {avat[i].tolist()}\n  oracle avatars {oa.tolist()}")
      return 1
    if glob_[i][1] != 0:
      print(f"step {s}: the episode ended on both sides (globals {glob_[i].tolist()}); equal so far")
  print(f"{steps} steps: equal after every step")
  return 0


if __name__ == "__main__":
  sys.exit(main())
