#!/usr/bin/env python3
"""Writes tests/golden/clean_up_modified_settings.pkl: the reference's own
`lab2d_settings` for clean_up (configs/substrates/clean_up.py build(roles, config),
7 players) as a plain dict tree — with an EDITED ASCII map, so that it is no
committed pack's map — plus the prefab overrides the tests apply to it at run time.
Runs where the reference tree is (this container); the GPU box has only the file.

  python tests/tools/make_settings_fixture.py
"""
import os
import pickle
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)

from meltingpot_amd import builder, refshim  # noqa: E402


def edited_map(ascii_map: str) -> str:
  """A sand bar across the river's left arm, two more orchard cells, one spawn point
  less: the object lists (dirt / water / apple sites), the state of every cell and
  the composite stacks of the renderer all differ from the committed pack's."""
  rows = ascii_map.strip("\n").split("\n")
  grid = [list(r) for r in rows]
  H, W = len(grid), len(grid[0])
  chars = sorted({c for r in grid for c in r})
  sand = " " if " " in chars else "."   # the map's plain floor character

  def first(ch):
    for y in range(H):
      for x in range(W):
        if grid[y][x] == ch:
          return y, x
    raise ValueError(ch)
  # 1. the first dirt-capable river cell becomes plain floor (one dirt site less)
  y, x = first("H")
  grid[y][x] = sand
  # 2. an orchard patch in the middle of the sand (four more apple sites)
  n = 0
  for yy in range(H):
    for xx in range(W - 1):
      if n < 4 and 8 <= yy <= 11 and grid[yy][xx] == sand and grid[yy][xx + 1] == sand and xx in (2, 3):
        grid[yy][xx] = "B"
        n += 1
  assert n == 4
  # 3. one spawn point less
  py, px = first("P")
  grid[py][px] = sand
  return "\n" + "\n".join("".join(r) for r in grid) + "\n"


def main():
  settings, module, config = refshim.build_settings("clean_up", ("default",) * 7)
  plain = builder._plain(settings)
  before = plain["simulation"]["map"]
  plain["simulation"]["map"] = edited_map(before)
  assert plain["simulation"]["map"] != before
  fixture = {
      "lab2d_settings": plain,
      "prefab_overrides": {
          "potential_apple": {"AppleGrow": {"maxAppleGrowthRate": 0.5,
                                            "thresholdDepletion": 0.9,
                                            "thresholdRestoration": 0.0}},
      },
      "source": "configs/substrates/clean_up.py build(('default',) * 7, get_config()), map edited "
                "by tests/tools/make_settings_fixture.py",
  }
  out = os.path.join(ROOT, "tests", "golden", "clean_up_modified_settings.pkl")
  with open(out, "wb") as f:
    pickle.dump(fixture, f, protocol=4)
  print(out, os.path.getsize(out), "bytes")
  print(plain["simulation"]["map"])


if __name__ == "__main__":
  main()
