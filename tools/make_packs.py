#!/usr/bin/env python3
"""Regenerate meltingpot_amd/assets/*.mpk from the reference configs.

Usage: python tools/make_packs.py [--reference /root/reference]

Runs the reference's own `configs/substrates/<name>.py:build()` (imported from
the reference tree through `meltingpot_amd.refshim`) and lowers the resulting
settings dict to an MPK1 pack (`meltingpot_amd/lower.py`).  The packs are
committed so that the GPU box (which has no reference tree) can run.
"""
import argparse
import random
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

from meltingpot_amd import lower, pack, refshim  # noqa: E402

TARGETS = {
    # pack name: (config module, number of players the pack is lowered for).
    # An engine runs any num_players <= that (MpConfig.num_players, = len(roles)
    # as in the reference): only the avatars differ with the player count.
    # clean_up: 15 = the avatar colours the config has (colors.human_readable
    # minus the one popped for Self); BASELINE.json runs 7 of them.
    "clean_up": ("clean_up", 15),
    # BASELINE.json configs[2]: 16 players (the reference default is 7)
    "commons_harvest__open": ("commons_harvest__open", 16),
    # same Lua level, walled-orchard map
    "commons_harvest__closed": ("commons_harvest__closed", 16),
    # same level, two-orchard map; Role / RoleBasedRewardTile are inert with the
    # default roles (lower.check_components)
    "commons_harvest__partnership": ("commons_harvest__partnership", 16),
    # BASELINE.json configs[3]: 9 players, TORUS map of 9 rooms (the resource
    # prefab has a claimed state, two paint sprites and two hits per player:
    # the reference's default count is the pack's maximum)
    "territory__rooms": ("territory__rooms", 9),
    # same Lua level on the 23 x 39 BOUNDED open map
    "territory__open": ("territory__open", 9),
    # per-episode 'choice' map characters (optional resources and spawn points)
    "territory__inside_out": ("territory__inside_out", 5),
    # coins.py draws the map size and the coin colours with Python's `random`
    # inside build(): the pack is the instance drawn after random.seed(0)
    "coins": ("coins", 2),
    # coop_mining (a sixth Lua level: ores, a mining beam): 8 = the byte an ore's miners
    # are kept in; the reference's default is 6.  Its two roles ("default", "target")
    # build identical avatars (agentRole "none" for all)
    "coop_mining": ("coop_mining", 8),
    # gift_refinements (a seventh Lua level: tokens, an inventory, a refining gift beam);
    # the reference's default is 6 players, both roles build the same avatar
    "gift_refinements": ("gift_refinements", 8),
}
# collaborative_cooking (an eighth Lua level), seven layouts, each lowered for its config's
# default roles (2 players; crowded 9, figure_eight 6)
for _layout in ("asymmetric", "circuit", "cramped", "crowded", "figure_eight", "forced", "ring"):
  TARGETS[f"collaborative_cooking__{_layout}"] = (f"collaborative_cooking__{_layout}", "default_roles")
# externality_mushrooms (a ninth Lua level: mushrooms whose rewards go to the eater, to everybody
# or to everybody else; spores, perishing, zapping with graduated sanctions and avatars that come
# back); its config has one role and five players
TARGETS["externality_mushrooms__dense"] = ("externality_mushrooms__dense", "default_roles")
# *_in_the_matrix (lua/levels/the_matrix): 2 players on the 15 x 23 maps
# (repeated, one_shot), 8 on the 24 x 25 arenas; lowered for the config's default
# roles (bach_or_stravinsky's two fan roles differ in Taste / DyadicRole kwargs,
# which the pack carries per player)
MATRIX_GAMES = ("prisoners_dilemma", "chicken", "stag_hunt", "pure_coordination",
                "rationalizable_coordination", "bach_or_stravinsky",
                "running_with_scissors")
for _game in MATRIX_GAMES:
  for _variant in ("repeated", "arena") + (("one_shot",) if _game == "running_with_scissors" else ()):
    TARGETS[f"{_game}_in_the_matrix__{_variant}"] = (
        f"{_game}_in_the_matrix__{_variant}", "default_roles")

# players an engine (and the oracle) runs when its caller names no count
# (MPK_HDR_DEFAULT_P), where that is not all the pack holds: BASELINE.json runs
# clean_up with the reference's 7.  The Substrate API always passes len(roles).
DEFAULT_PLAYERS = {"clean_up": 7, "coop_mining": 6, "gift_refinements": 6}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--reference", default=refshim.DEFAULT_REFERENCE_ROOT)
  ap.add_argument("--out", default=os.path.join(
      os.path.dirname(__file__), "..", "meltingpot_amd", "assets"))
  ap.add_argument("--only", nargs="*", default=[],
                  help="only the packs whose name contains one of these strings")
  args = ap.parse_args()
  os.makedirs(args.out, exist_ok=True)
  for pack_name, (module, players) in TARGETS.items():
    if args.only and not any(s in pack_name for s in args.only):
      continue
    random.seed(0)
    if players == "default_roles":
      roles = tuple(refshim.load_config_module(module, args.reference)
                    .get_config().default_player_roles)
    else:
      roles = ("default",) * players
    settings, mod, config = refshim.build_settings(module, roles, args.reference)
    action_set = getattr(mod, "ACTION_SET", None)
    if action_set is None and module.startswith("collaborative_cooking__"):
      action_set = sys.modules["meltingpot.configs.substrates.collaborative_cooking"].ACTION_SET
    if action_set is None and module.startswith("externality_mushrooms__"):
      action_set = sys.modules["meltingpot.configs.substrates.externality_mushrooms"].ACTION_SET
    if action_set is None:  # territory__rooms re-uses its base config's table
      action_set = sys.modules["meltingpot.configs.substrates.territory"].ACTION_SET
    if pack_name == "coins":
      # coins.py draws the map size inside build() (get_ascii_map, :45-82): every
      # environment has its own map, for all its episodes.  The pack holds all 36
      # (width, height) maps as the outcomes of one per-world choice (the coin
      # colours stay those of this instance: no rule depends on them).
      settings = lower.coins_with_every_map(settings, mod, config)
    tables = lower.lower(module, settings, action_set,
                         default_players=DEFAULT_PLAYERS.get(pack_name, 0))
    valid = sorted(config.valid_roles) if hasattr(config, "valid_roles") else ["default"]
    if len(valid) > 1 and "mx_player_i32" not in tables:
      # roles that change nothing the pack holds (coop_mining): checked, not stored
      for role in valid:
        random.seed(0)
        s2, _, _ = refshim.build_settings(module, (role,) * len(roles), args.reference)
        t2 = lower.lower(module, s2, action_set, default_players=DEFAULT_PLAYERS.get(pack_name, 0))
        assert pack.dumps(t2) == pack.dumps(tables), f"role {role!r} changes the pack"
    elif len(valid) > 1:
      # per-player constants of every role (bach_or_stravinsky: row / column player
      # and avatar colour by role), so that any assignment can be created
      per_role = {}
      for role in valid:
        random.seed(0)
        s2, _, _ = refshim.build_settings(module, (role,) * len(roles), args.reference)
        per_role[role] = lower.lower(module, s2, action_set,
                                     default_players=DEFAULT_PLAYERS.get(pack_name, 0))
      lower.add_role_tables(tables, roles, per_role)
    blob = pack.dumps(tables)
    path = os.path.join(args.out, f"{pack_name}.mpk")
    with open(path, "wb") as f:
      f.write(blob)
    print(f"{path}: {len(blob)} bytes, {len(tables)} tables")


if __name__ == "__main__":
  main()
