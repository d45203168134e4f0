#!/usr/bin/env python3
"""Replays a recorded trajectory through the CPU oracle under every combination
of its engine-assumption switches and reports where each first diverges.

This is the fitting half of SURVEY.md section 8(f) rank 4: `tools/
dump_dmlab2d_trace.py` records a DMLab2D trajectory (needs the dmlab2d wheel);
this tool takes such a file, drives the oracle with the recorded actions and
compares rewards and pixels frame by frame.  A switch combination that
reproduces the trace where the others do not is the engine's behaviour.

  python tests/tools/replay_trace.py trace.npz [--substrate clean_up] [--players 7]

Trace format (`np.savez`, as dump_dmlab2d_trace.py writes it):
  actions     int   [T, P]                discrete ids into the ACTION_SET
  rewards     f64   [T, P]
  world_rgb   u8    [T + 1, H*8, W*8, 3]  frame 0 = after reset
  rgb         u8    [T + 1, P, 88, 88, 3]
  seed        int                         env_seed of the recording
  position    int   [T + 1, P, 2]         optional (debug observations): x, y
  orientation int   [T + 1, P]            optional
  alive       u8    [T + 1, P]            optional (derived from black frames otherwise)

What can and cannot be compared.  The reference's generator is mt19937_64, the
oracle's is counter-based (A10), so nothing that depends on a draw can match a
DMLab2D recording: spawn points, respawn points, growth, animation phases, the
visiting order of a frame (A1).  Two mechanisms deal with that:
  * teacher forcing — with `position` / `orientation` in the trace, the avatars
    are put where the recording has them before every step, so each frame is a
    one-step prediction from the recorded state and an unlucky draw costs one
    frame, not the rest of the episode;
  * masks — pixels of cells that hold a piece with a random start phase or a
    probabilistic updater (water, apples, dirt, resources, coins) are left out
    of the pixel comparison.
A trace the oracle itself produced (same generator) needs neither: the matching
combination reproduces it exactly, which is what the CPU test uses.
"""
import argparse
import dataclasses
import itertools
import os
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

# the boolean switches a trace can decide, with the values to try
SWITCHES: Dict[str, Tuple[int, ...]] = {
    "A1_shuffle_order": (1, 0),
    "A2_flush_count": (128, 1),
    "A3b_blocked_move_reenters": (1, 0),
    "A4_beam_marks_blocked": (1, 0),
    "A5_teleport_free_only": (0, 1),
    "A6_dead_view_black": (1, 0),
    # the generator: counter-based (A10) or ONE serial mt19937_64 in call order (A10s)
    "A10s_serial_mt19937": (0, 1),
}

# object kinds (include/mp_pack.h MPK_KIND_*) whose look depends on a draw
_RANDOM_KINDS = {16, 17, 18, 19, 20, 22, 26}


@dataclasses.dataclass
class Divergence:
  frame: int
  what: str          # "reward", "WORLD.RGB", "<p>.RGB"
  where: Tuple[int, ...]


@dataclasses.dataclass
class Report:
  switches: Dict[str, int]
  frames: int
  first: Optional[Divergence]
  bad_frames: int            # frames with any divergence
  first_by_observation: Dict[str, Divergence]

  def line(self) -> str:
    sw = " ".join(f"{k.split('_')[0]}={v}" for k, v in self.switches.items())
    if self.first is None:
      return f"{sw}: reproduces all {self.frames} frames"
    f = self.first
    return (f"{sw}: {self.bad_frames} of {self.frames} frames differ, first at frame "
            f"{f.frame} in {f.what} {f.where}")


def load_trace(path: str) -> Dict[str, np.ndarray]:
  with np.load(path) as z:
    return {k: z[k] for k in z.files}


def random_cell_mask(tables) -> np.ndarray:
  """[H, W] bool: cells holding a piece whose look depends on a draw."""
  hdr = tables["hdr"]
  H, W = int(hdr[2]), int(hdr[3])
  mask = np.zeros((H, W), bool)
  for kind, x, y, _ in np.asarray(tables["objects"]).reshape(-1, 4):
    if int(kind) in _RANDOM_KINDS:
      mask[y, x] = True
  return mask


def _first_diff(a: np.ndarray, b: np.ndarray, keep: Optional[np.ndarray]):
  d = (a != b)
  if d.ndim == 3:
    d = d.any(-1)
  if keep is not None:
    d &= keep
  if not d.any():
    return None
  return tuple(int(v) for v in np.argwhere(d)[0])


def replay(pack_bytes: bytes, trace: Dict[str, np.ndarray], switches: Dict[str, int],
           players: int = 0, mask_random_cells: bool = True,
           max_frames: Optional[int] = None) -> Report:
  from meltingpot_amd import pack as pack_lib
  from oracle import oracle as oracle_lib
  actions = np.asarray(trace["actions"], np.int32)
  T, P = actions.shape
  if max_frames:
    T = min(T, max_frames)
  o = oracle_lib.Oracle(pack_bytes, int(trace["seed"]), players or P)
  assert o.P == P, (o.P, P)
  for k, v in switches.items():
    o.set_option(k, v)
  o.reset()
  forced = "position" in trace and "orientation" in trace
  alive = trace.get("alive")

  def force(t):
    for p in range(P):
      a = bool(alive[t, p]) if alive is not None else bool(trace["rgb"][t, p].any())
      x, y = trace["position"][t, p]
      o.place_avatar(p, x, y, trace["orientation"][t, p], a)

  tables = pack_lib.loads(pack_bytes)
  cell_keep = ~random_cell_mask(tables) if mask_random_cells else None
  world_keep = None
  if cell_keep is not None:
    world_keep = np.kron(cell_keep, np.ones((8, 8), bool))

  first: Optional[Divergence] = None
  first_by: Dict[str, Divergence] = {}
  bad = 0

  def note(frame, what, where):
    nonlocal first
    d = Divergence(frame, what, where)
    first_by.setdefault(what, d)
    if first is None:
      first = d

  def compare(t):
    nonlocal bad
    hit = False
    w = _first_diff(o.render_world(), trace["world_rgb"][t], world_keep)
    if w is not None:
      note(t, "WORLD.RGB", w); hit = True
    if "rgb" in trace and cell_keep is None:   # agent views: only unmasked comparison
      for p in range(P):
        w = _first_diff(o.render_agent(p), trace["rgb"][t, p], None)
        if w is not None:
          note(t, f"{p + 1}.RGB", w); hit = True
    if t > 0:
      r = np.flatnonzero(o.rewards() != trace["rewards"][t - 1])
      if r.size and cell_keep is None:   # rewards follow draws (apples, resources): unmasked only
        note(t, "reward", (int(r[0]),)); hit = True
    bad += hit

  if forced:
    force(0)
  compare(0)
  for t in range(T):
    if forced:
      force(t)
    o.step(actions[t])
    compare(t + 1)
  o.close()
  return Report(dict(switches), T + 1, first, bad, first_by)


def fit(pack_bytes: bytes, trace: Dict[str, np.ndarray], players: int = 0,
        names: Optional[Sequence[str]] = None, **kw) -> List[Report]:
  """Replays under every combination of the switches in `names`; best first
  (fewest diverging frames, then latest first divergence)."""
  names = list(names or SWITCHES)
  reports = []
  for values in itertools.product(*(SWITCHES[n] for n in names)):
    reports.append(replay(pack_bytes, trace, dict(zip(names, values)), players, **kw))
  reports.sort(key=lambda r: (r.bad_frames, -(r.first.frame if r.first else 1 << 30)))
  return reports


def record_with_oracle(pack_bytes: bytes, seed: int, actions: np.ndarray,
                       switches: Optional[Dict[str, int]] = None,
                       players: int = 0) -> Dict[str, np.ndarray]:
  """A trace in the dump_dmlab2d_trace.py format, produced by the oracle itself
  (the synthetic input of the CPU test; also documents the format)."""
  from oracle import oracle as oracle_lib
  T, P = actions.shape
  o = oracle_lib.Oracle(pack_bytes, seed, players or P)
  for k, v in (switches or {}).items():
    o.set_option(k, v)
  o.reset()
  world, rgb, rew, pos, ori, alive = [], [], [], [], [], []

  def snap():
    world.append(o.render_world())
    rgb.append(np.stack([o.render_agent(p) for p in range(P)]))
    _, avat, _ = o.dump()
    pos.append(avat[:, :2].copy()); ori.append(avat[:, 2].copy()); alive.append(avat[:, 3].copy())

  snap()
  for t in range(T):
    o.step(actions[t])
    rew.append(o.rewards().copy())
    snap()
  o.close()
  return {"actions": np.asarray(actions, np.int32), "rewards": np.array(rew),
          "world_rgb": np.array(world), "rgb": np.array(rgb), "seed": np.int64(seed),
          "position": np.array(pos), "orientation": np.array(ori),
          "alive": np.array(alive, np.uint8)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("trace")
  ap.add_argument("--substrate", default="clean_up")
  ap.add_argument("--players", type=int, default=0)
  ap.add_argument("--frames", type=int, default=0)
  ap.add_argument("--no-mask", action="store_true",
                  help="compare every pixel and the rewards (a trace recorded with the "
                       "oracle's own generator)")
  args = ap.parse_args()
  from meltingpot_amd import engine
  reports = fit(engine.load_pack(args.substrate), load_trace(args.trace), args.players,
                mask_random_cells=not args.no_mask, max_frames=args.frames or None)
  for r in reports:
    print(r.line())


if __name__ == "__main__":
  main()
