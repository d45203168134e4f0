#!/usr/bin/env python3
"""Writes the fixtures of tests/test_reference_frames.py — the only frames of a REAL DMLab2D
run the reference tree holds for a level built from the components this engine restates:

  tests/golden/tutorial_harvest_settings.pkl   the lab2d settings of the tutorial level
      `harvest_finished` (examples/tutorial/harvest/configs/environment/harvest_finished.py:
      22 x 11 cells, walls, apples, five avatars with a 3 / 3 / 5 / 1 view), as a plain dict
      tree, with what the reference's builder adds before dmlab2d sees them:
        * `playerPalettes`: empty in the config; builder.py:116-125 /
          game_object_utils.py then takes the first five colours of utils/substrates/colors.py
          through shapes.get_palette — computed HERE with the reference's own modules;
        * a `scene` object and `gameObjects` (base_simulation.lua's defaults);
        * the state `playerWait` (layer-less) on the avatar prefab: its Avatar component
          names it as waitState and its StateManager never declares it — dmlab2d only meets
          it when an avatar dies;
        * the map text WITHOUT its last line: today's config ends the text with the
          indentation of its closing quotes ("\n  "), which dmlab2d makes a twelfth row of
          empty cells (as it does with coins' padding rows of blanks: WORLD.RGB is 136 x 136
          there whatever size was drawn, configs/substrates/coins.py:45-83,483) — the
          recording shows eleven (640 x 320 for 22 x 11 cells): it predates that line.
  tests/golden/tutorial_harvest_frames.npz     frame 0 of docs/substrate_tutorial/images/
      harvest.gif (WORLD.RGB, 176 x 88 shown at 640 x 320) and of playerview.gif (one
      player's RGB, 56 x 56 shown at 640 x 640), decoded to uint8 RGB.  The GIFs are lossy
      (a shared palette, dithering): the test compares within that noise.  And `player_track`:
      ALL 343 frames of playerview.gif brought back to the observation's own 56 x 56 (the mean
      of the window pixels that show each observation pixel), uint8.

Runs where the reference tree is (this container); the GPU box has only the files.

  python tests/tools/make_tutorial_frames_fixture.py
"""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)

from meltingpot_amd import builder, refshim  # noqa: E402

REF = refshim.DEFAULT_REFERENCE_ROOT


def main():
  from PIL import Image, ImageSequence
  refshim._install_stubs(REF)
  mod = refshim._load(os.path.join(REF, "examples", "tutorial", "harvest", "configs", "environment",
                                   "harvest_finished.py"), "tutorial_harvest_finished")
  settings = builder._plain(mod.get_config().lab2d_settings)
  colors = sys.modules["meltingpot.utils.substrates.colors"]
  shapes = sys.modules["meltingpot.utils.substrates.shapes"]
  sim = settings["simulation"]
  assert sim["playerPalettes"] == [] and int(settings["numPlayers"]) == 5
  sim["playerPalettes"] = [dict(shapes.get_palette(colors.palette[i])) for i in range(5)]
  sim.setdefault("gameObjects", [])
  sim.setdefault("scene", {"name": "scene", "components": [
      {"component": "StateManager",
       "kwargs": {"initialState": "scene", "stateConfigs": [{"state": "scene"}]}},
      {"component": "Transform"}]})
  states = sim["prefabs"]["avatar"]["components"][0]["kwargs"]["stateConfigs"]
  assert [s["state"] for s in states] == ["player"]
  states.append({"state": "playerWait"})
  rows = sim["map"].split("\n")
  assert rows[-1] == "  " and set(rows[-2]) == {"*"}
  sim["map"] = "\n".join(rows[:-1]) + "\n"
  out = os.path.join(ROOT, "tests", "golden", "tutorial_harvest_settings.pkl")
  with open(out, "wb") as f:
    pickle.dump({"lab2d_settings": settings,
                 "source": "examples/tutorial/harvest/configs/environment/harvest_finished.py "
                           "get_config().lab2d_settings + the builder's default palettes "
                           "(tests/tools/make_tutorial_frames_fixture.py)"}, f, protocol=4)
  print(out, os.path.getsize(out), "bytes")
  frames = {}
  for key, name in (("world", "harvest.gif"), ("player", "playerview.gif")):
    gif = Image.open(os.path.join(REF, "docs", "substrate_tutorial", "images", name))
    frames[key] = np.array(next(ImageSequence.Iterator(gif)).convert("RGB"), np.uint8)
    print(name, frames[key].shape)
  gif = Image.open(os.path.join(REF, "docs", "substrate_tutorial", "images", "playerview.gif"))
  edges = [round(i * 640 / 56) for i in range(57)]
  count = (np.diff(edges)[:, None] * np.diff(edges)[None, :]).astype(np.float32)
  track = []
  for fr in ImageSequence.Iterator(gif):
    img = np.array(fr.convert("RGB")).astype(np.float32)
    block = np.add.reduceat(np.add.reduceat(img, edges[:-1], axis=0), edges[:-1], axis=1)
    track.append(np.clip(np.rint(block / count[:, :, None]), 0, 255).astype(np.uint8))
  frames["player_track"] = np.stack(track)
  print("playerview.gif:", frames["player_track"].shape)
  out = os.path.join(ROOT, "tests", "golden", "tutorial_harvest_frames.npz")
  np.savez_compressed(out, **frames)
  print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
  main()
