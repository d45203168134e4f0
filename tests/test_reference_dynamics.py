"""Rules against what HAPPENS in a recording of a real DMLab2D run.

`docs/substrate_tutorial/images/harvest.gif` (see tests/test_reference_frames.py for its first
frame) is 461 frames of the reference's tutorial level being played on dmlab2d: five avatars —
the human steers one at a time —, sixty apples that are eaten and regrow.  Reduced to the cell
of every avatar and the set of visible apples per frame (`tests/golden/
tutorial_harvest_recording.json`, `tests/tools/make_tutorial_dynamics_fixture.py`) it is the one
place in the reference tree where the ENGINE CYCLE can be watched: 85 moves, 40 of them onto
an apple, 29 regrowths.

1. What the recording says about dmlab2d + the library components (no code of this repo
   involved): an avatar moves one cell a frame and never onto a wall or another avatar; an
   apple disappears IN THE FRAME an avatar arrives on it — `Edible:onEnter`'s setState is
   queued by a contact callback of the move and still lands inside the same `grid:update`
   (events queued by callbacks run in a later flush of the SAME update) — and
   never otherwise; an apple comes back only next to a live one (the tutorial's DensityRegrow:
   rate = live neighbours in the diamond of radius 1 x baseRate) and never under an avatar.
2. The same recording replayed on THIS repo's rules: the oracle of commons_harvest — the level
   whose `Avatar` (avatar_library.lua) and `Edible` (component_library.lua:953-1004) are the
   components the tutorial level uses — on the tutorial's map, regrowth switched off (it is
   random: every regrowth of the recording starts a fresh segment from the recorded state),
   every avatar driven by the move its recorded step implies: after EVERY step the avatars
   stand where the recording has them, the apples that are left are the recording's, and the
   eater — nobody else — was paid 1.0 in that very step."""
import json
import os
import pickle

import numpy as np
import pytest

from meltingpot_amd import builder, lower, pack
from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NOOP, FORWARD, BACKWARD, STEP_LEFT, STEP_RIGHT = 0, 1, 2, 3, 4
MOVE = {(0, -1): FORWARD, (0, 1): BACKWARD, (-1, 0): STEP_LEFT, (1, 0): STEP_RIGHT}   # facing north


@pytest.fixture(scope="module")
def recording():
  with open(os.path.join(GOLDEN, "tutorial_harvest_recording.json")) as f:
    rec = json.load(f)
  sites = [tuple(s) for s in rec["apple_sites"]]
  frames = [([tuple(c) for c in fr["avatars"]], {sites[i] for i in fr["apples"]}) for fr in rec["frames"]]
  return rec["map"], sites, frames


def test_what_the_recording_says_about_the_engine(recording):
  rows, sites, frames = recording
  assert len(frames) == 461 and len(sites) == 60 and frames[0][1] == set(sites)
  walls = {(x, y) for y, r in enumerate(rows) for x, ch in enumerate(r) if ch == "*"}
  moves = eaten = regrown = 0
  for t in range(1, len(frames)):
    (was, apples_before), (now, apples_now) = frames[t - 1], frames[t]
    assert len(set(now)) == 5 and not set(now) & walls
    movers = [p for p in range(5) if now[p] != was[p]]
    assert len(movers) <= 1                              # the human steers one avatar at a time
    for p in movers:
      step = abs(now[p][0] - was[p][0]) + abs(now[p][1] - was[p][1])
      assert step == 1 or (t == 214 and step == 2)       # (one frame of the recording is missing)
      moves += 1
    arrived = {now[p] for p in movers}
    if t == 214:
      arrived.add((7, 8))                                # (the cell passed in the missing frame)
    gone, new = apples_before - apples_now, apples_now - apples_before
    assert gone == arrived & apples_before, (t, gone, arrived)      # eaten on arrival, in that frame
    eaten += len(gone)
    for (x, y) in new:
      assert (x, y) not in was and (x, y) not in now
      live_neighbours = {(x + 1, y), (x - 1, y), (x, y + 1), (x, y - 1)} & apples_before
      assert live_neighbours, (t, (x, y))
      regrown += 1
  assert (moves, eaten, regrown) == (85, 40, 29)


def _segment_oracle(settings, action_set, rows, sites, apples, cells):
  s = pickle.loads(pickle.dumps(settings))
  text = [list(r.replace("*", "W")) for r in rows]
  # (commons_harvest__open's first two avatars spawn from a group of their own, 'Q':
  # commons_harvest__open.py:515-528 — wherever reset() drops them, they are placed below)
  spawn = [(x, y) for y, r in enumerate(text) for x, ch in enumerate(r) if ch == "_"]
  for i, (x, y) in enumerate(spawn):
    text[y][x] = "Q" if i < 2 else "P"
  for (x, y) in sites:
    text[y][x] = "A" if (x, y) in apples else "G"
  s["simulation"]["map"] = "\n" + "\n".join("".join(r) for r in text) + "\n"
  overrides = {"apple": {"DensityRegrow": {"regrowthProbabilities": [0.0, 0.0, 0.0, 0.0]}}}
  _, blob, _ = builder.lower_settings(s, overrides, action_set=action_set)
  o = oracle.Oracle(blob, 7, 5)
  o.reset()
  for p in range(5):
    o.place_avatar(p, 1 + p, 1, 0, alive=False)
  for p, (x, y) in enumerate(cells):
    assert o.place_avatar(p, x, y, 0)                    # facing north: a move IS its compass direction
  return o, pack.loads(blob)


def test_the_recording_replayed_on_this_repos_rules(recording):
  rows, sites, frames = recording
  with open(os.path.join(GOLDEN, "tutorial_on_commons_settings.pkl"), "rb") as f:
    fixture = pickle.load(f)
  settings, action_set = fixture["lab2d_settings"], fixture["action_set"]
  assert [a["move"] for a in action_set[:5]] == [0, 1, 3, 4, 2]    # NOOP FORWARD BACKWARD LEFT RIGHT
  o = tables = None
  segments = paid = 0
  for t in range(1, len(frames)):
    (was, apples_before), (now, apples_now) = frames[t - 1], frames[t]
    if o is None:
      o, tables = _segment_oracle(settings, action_set, rows, sites, apples_before, was)
      live = int(tables["ch_states"][0])
      apple_layer = int(tables["state_layer"][live])
      segments += 1
    # the steps this frame implies (two for the frame the recording lost)
    plans = [[NOOP] * 5]
    for p in range(5):
      dx, dy = now[p][0] - was[p][0], now[p][1] - was[p][1]
      if (dx, dy) in MOVE:
        plans[0][p] = MOVE[(dx, dy)]
      elif (dx, dy) != (0, 0):
        assert t == 214 and (dx, dy) == (0, -2)
        plans[0][p] = FORWARD
        plans.append([FORWARD if q == p else NOOP for q in range(5)])
    eaters = {}
    for acts in plans:
      o.step(np.asarray(acts, np.int32))
      for p, r in enumerate(o.rewards()):
        if r:
          eaters[p] = eaters.get(p, 0.0) + float(r)
    grid, avat, _ = o.dump()
    assert [(int(a[0]), int(a[1])) for a in avat] == list(now), t
    ours = {(x, y) for (x, y) in sites if grid[apple_layer, y, x] == live}
    regrown = apples_now - apples_before
    assert ours == apples_now - regrown, (t, ours ^ (apples_now - regrown))
    # paid in the step of the arrival, 1.0 an apple, nobody else
    gone = apples_before - apples_now
    want = {p: float(len(gone)) for p in range(5) if now[p] != was[p] and gone}
    assert eaters == want, (t, eaters, want)
    paid += len(gone)
    if regrown:                                          # random in the reference: a fresh segment
      o.close()
      o = None
  assert segments == 27 and paid == 40
