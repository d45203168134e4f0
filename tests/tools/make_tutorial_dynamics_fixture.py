#!/usr/bin/env python3
"""Writes tests/golden/tutorial_harvest_recording.json and tutorial_on_commons_settings.pkl:
what HAPPENS in the reference's recording of its tutorial level on a real DMLab2D
(docs/substrate_tutorial/images/harvest.gif: 461 frames of WORLD.RGB, one env step a frame but
for one dropped frame), reduced to what a rule test needs —

  * per frame, the cell of each of the five avatars (a cell of the frame holds avatar p when a
    quarter of its inner pixels are within 40 of p's default colour, colors.palette[p]) and the
    set of apple sites that show an apple (a fifth of the inner pixels green-dominant; a cell an
    avatar stands on never counts: the second player IS green-dominant);
  * the lab2d settings of commons_harvest__open (the reference's own build(), five players) with
    the TUTORIAL's map in place of its own ('*' -> 'W', '_' -> 'P') and a map character "G" for
    a site whose apple is gone (grass only): the level whose Avatar (avatar_library.lua) and
    Edible (component_library.lua:953-1004) components are the ones the tutorial level uses
    too — what tests/test_reference_dynamics.py replays the recording on.

  python tests/tools/make_tutorial_dynamics_fixture.py      (needs /root/reference and PIL)
"""
import json
import os
import pickle
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)

from meltingpot_amd import builder, refshim  # noqa: E402

REF = refshim.DEFAULT_REFERENCE_ROOT
W, H = 22, 11


def inner(img, x, y):
  h, w = img.shape[:2]
  return img[round(y * h / H) + 3:round((y + 1) * h / H) - 3,
             round(x * w / W) + 3:round((x + 1) * w / W) - 3].reshape(-1, 3)


def main():
  from PIL import Image, ImageSequence
  refshim._install_stubs(REF)
  colors = sys.modules["meltingpot.utils.substrates.colors"]
  palette = [np.array(colors.palette[p][:3], np.int32) for p in range(5)]
  with open(os.path.join(ROOT, "tests", "golden", "tutorial_harvest_settings.pkl"), "rb") as f:
    rows = pickle.load(f)["lab2d_settings"]["simulation"]["map"].strip("\n").split("\n")
  assert len(rows) == H and all(len(r) == W for r in rows)
  sites = [(x, y) for y, r in enumerate(rows) for x, ch in enumerate(r) if ch == "A"]
  gif = Image.open(os.path.join(REF, "docs", "substrate_tutorial", "images", "harvest.gif"))
  frames = []
  for fr in ImageSequence.Iterator(gif):
    img = np.array(fr.convert("RGB")).astype(np.int32)
    avatars, apples = {}, []
    for y in range(1, H - 1):
      for x in range(1, W - 1):
        px = inner(img, x, y)
        for p, c in enumerate(palette):
          if (np.abs(px - c).max(axis=1) < 40).mean() > 0.25:
            avatars.setdefault(p, []).append([x, y])
        if (x, y) in sites and ((px[:, 1] > px[:, 0] + 40) & (px[:, 1] > px[:, 2] + 40)).mean() > 0.2:
          apples.append(sites.index((x, y)))
    assert sorted(avatars) == list(range(5)) and all(len(v) == 1 for v in avatars.values()), len(frames)
    cells = [avatars[p][0] for p in range(5)]
    apples = [i for i in apples if list(sites[i]) not in cells]
    frames.append({"avatars": cells, "apples": apples})
  out = os.path.join(ROOT, "tests", "golden", "tutorial_harvest_recording.json")
  with open(out, "w") as f:
    json.dump({"source": "docs/substrate_tutorial/images/harvest.gif, reduced by "
                         "tests/tools/make_tutorial_dynamics_fixture.py",
               "map": rows, "apple_sites": [list(s) for s in sites], "frames": frames}, f,
              separators=(",", ":"))
  print(out, os.path.getsize(out), "bytes;", len(frames), "frames")

  settings, module, _ = refshim.build_settings("commons_harvest__open", ("default",) * 5)
  plain = builder._plain(settings)
  sim = plain["simulation"]
  assert set(sim["charPrefabMap"]) >= {"P", " ", "W", "A"} and "grass" in sim["prefabs"]
  sim["charPrefabMap"]["G"] = "grass"
  sim["map"] = "\n" + "\n".join(r.replace("*", "W").replace("_", "P") for r in rows) + "\n"
  out = os.path.join(ROOT, "tests", "golden", "tutorial_on_commons_settings.pkl")
  with open(out, "wb") as f:
    pickle.dump({"lab2d_settings": plain,
                 "action_set": [dict(a) for a in module.ACTION_SET],
                 "source": "configs/substrates/commons_harvest__open.py build(('default',) * 5) "
                           "with the tutorial level's map (tests/tools/make_tutorial_dynamics_fixture.py)"},
                f, protocol=4)
  print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
  main()
