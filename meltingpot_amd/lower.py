"""Lowering: reference lab2d settings dict -> flat constant tables (an MPK1 pack).

Input is exactly what the reference hands to `builder.builder()`
(`meltingpot/utils/substrates/builder.py:142`), i.e. the dict returned by
`configs/substrates/<name>.py:build()`.  Output is the numeric form of what the
reference's Lua side derives from it at `api:init`
(`lua/modules/api_factory.lua:53-67`, `base_simulation.lua:77-148,253-345`):

  * object list in creation order: scene, avatars, then map objects row-major
    with `all` lists expanded in order (`base_simulation.lua:103-131`,
    `prefab_utils.lua:59-109`);
  * state table: state -> (layer, sprite, groups, contact)
    (`component_library.lua:90-109`);
  * render order: the 7 base layers then hit layers in registration order
    (`base_simulation.lua:263-271`, `avatar_library.lua:597-603`,
    `clean_up/components.lua:185-190`);
  * sprite atlas: every sprite as 4 facings x 8x8 RGBA
    (`component_library.lua:567-604`), 16x16 art box-averaged to spriteSize
    (assumption A8);
  * per-substrate rule constants (component kwargs) and site lists.

Nothing here runs per step; the engine and the oracle both consume the pack.
"""

from __future__ import annotations

import math
from fractions import Fraction
from typing import Any, Dict, List, Mapping, Sequence, Tuple

import numpy as np

COMPASS = ("N", "E", "S", "W")
BASE_RENDER_ORDER = ("logic", "alternateLogic", "background", "lowerPhysical",
                     "upperPhysical", "overlay", "superOverlay")

SUBSTRATE_IDS = {"clean_up": 1, "commons_harvest": 2, "territory": 3, "coins": 4,
                 "the_matrix": 5, "coop_mining": 6, "gift_refinements": 7,
                 "collaborative_cooking": 8, "externality_mushrooms": 9}

# Object kinds (by the rule-bearing component an object carries).
KIND_SCENE, KIND_AVATAR, KIND_STATIC = 0, 1, 2
KIND_APPLE_GROW, KIND_DIRT, KIND_ANIM = 16, 17, 18
KIND_DENSITY_REGROW, KIND_RESOURCE, KIND_OVERLAY = 19, 20, 21
KIND_REWARD_INDICATOR, KIND_TEXTURE, KIND_DAMAGE_INDICATOR, KIND_MARKING = 22, 23, 24, 25
KIND_COIN = 26
KIND_READY_MARKER = 27
KIND_ORE = 28
KIND_TOKEN = 29
KIND_CONTAINER, KIND_RECEIVER, KIND_POT, KIND_INVENTORY, KIND_LOADING_BAR = 8, 9, 10, 11, 12
KIND_MUSHROOM = 13

HDR_VERSION, HDR_SUBSTRATE, HDR_H, HDR_W, HDR_L, HDR_NSTATES, HDR_NSPRITES, \
    HDR_P, HDR_SPRITE, HDR_TOPOLOGY, HDR_VL, HDR_VR, HDR_VF, HDR_VB, \
    HDR_MAXFRAMES, HDR_NOBJ, HDR_NACT, HDR_NGROUPS, HDR_AVATAR_LAYER, \
    HDR_NHITS = range(20)
# players an engine runs when the caller names no count (0 = all the pack holds)
HDR_DEFAULT_P = 20
HDR_NFIELDS = 21   # raw action fields per avatar (len(actionOrder) <= 4)
HDR_LEN = 64

SPRITE_FLAG_PARTIAL = 1   # some pixel has 0 < alpha < 255
SPRITE_FLAG_OPAQUE = 2    # every pixel of every facing has alpha 255
SPRITE_FLAG_EMPTY = 4     # every pixel has alpha 0


def prob_threshold(p: float) -> int:
  """Integer T such that for r uniform in [0, 2^53):  r * 2^-53 < p  <=>  r < T.

  The reference compares `random:uniformReal(0, 1) < p` in doubles
  (`clean_up/components.lua:77`); with a 53-bit uniform the comparison is an
  exact integer compare against ceil(p * 2^53).
  """
  if not p > 0.0:
    return 0
  t = math.ceil(Fraction(p) * (1 << 53))
  return min(t, 1 << 53)


def _get_component(obj: Mapping[str, Any], name: str):
  for c in obj["components"]:
    if c["component"] == name:
      return c
  return None


def _components(obj, name):
  return [c for c in obj["components"] if c["component"] == name]


def _text_to_rgba(text: str, palette: Mapping[str, Sequence[int]]) -> np.ndarray:
  # sprite art may be indented inside a Python function (territory.py:512-521)
  lines = [ln.strip() for ln in text.strip().split("\n")]
  h = len(lines)
  w = len(lines[0])
  img = np.zeros((h, w, 4), np.uint8)
  for y, ln in enumerate(lines):
    assert len(ln) == w, "ragged sprite"
    for x, ch in enumerate(ln):
      c = palette[ch]
      img[y, x, :3] = c[:3]
      img[y, x, 3] = c[3] if len(c) > 3 else 255
  return img


def _fit(img: np.ndarray, size: int) -> np.ndarray:
  """Assumption A8: integer box average with round-half-up when the art is an
  integer multiple of spriteSize (clean_up water is 16x16 at spriteSize 8,
  `shapes.py:1115`, `clean_up.py:531-533,855`)."""
  h, w, _ = img.shape
  if h == size and w == size:
    return img
  assert h % size == 0 and w % size == 0 and h // size == w // size, (h, w)
  k = h // size
  acc = img.astype(np.uint32).reshape(size, k, size, k, 4).sum(axis=(1, 3))
  return ((acc + (k * k) // 2) // (k * k)).astype(np.uint8)


def _solid(color: Sequence[int], size: int) -> np.ndarray:
  img = np.zeros((size, size, 4), np.uint8)
  img[..., :3] = color[:3]
  img[..., 3] = color[3] if len(color) > 3 else 255
  return img


class _Sprites:
  """Sprite registry: name -> 4 facings (N,E,S,W) of size x size RGBA."""

  def __init__(self, size: int):
    self.size = size
    self.names: List[str] = []
    self.images: List[np.ndarray] = []

  def index(self, name: str) -> int:
    return self.names.index(name)

  def _slot(self, name: str) -> int:
    if name in self.names:
      return self.names.index(name)
    self.names.append(name)
    self.images.append(np.zeros((4, self.size, self.size, 4), np.uint8))
    return len(self.names) - 1

  def add_color(self, name: str, color: Sequence[int]) -> None:
    i = self._slot(name)
    self.images[i][:] = _solid(color, self.size)

  def add_shape(self, name: str, text, palette, no_rotate: bool) -> None:
    # component_library.lua:567-604 (_addShapesToTileSet)
    i = self._slot(name)
    if isinstance(text, (list, tuple)) and len(text) == 4:
      assert no_rotate
      for j in range(4):
        self.images[i][j] = _fit(_text_to_rgba(text[j], palette), self.size)
      return
    img = _fit(_text_to_rgba(text, palette), self.size)
    for j in range(4):
      # Assumption A9: facing j = art rotated 90 deg clockwise j times.
      self.images[i][j] = img if no_rotate else np.rot90(img, -j)

  def add_from_appearance(self, kw: Mapping[str, Any], custom: bool = False) -> None:
    key = ((lambda s: "custom" + s[0].upper() + s[1:]) if custom
           else (lambda s: s))
    names = kw.get(key("spriteNames"), [])
    mode = kw.get("renderMode", "colored_square")
    colors = kw.get(key("spriteRGBColors"), [])
    shapes = kw.get(key("spriteShapes"), [])
    palettes = kw.get(key("palettes"), [])
    no_rot = kw.get(key("noRotates"), [])
    for i, name in enumerate(names):
      if mode == "colored_square":
        self.add_color(name, colors[i])
      elif mode == "ascii_shape":
        nr = bool(no_rot[i]) if i < len(no_rot) else False
        self.add_shape(name, shapes[i], palettes[i], nr)

  def arrays(self) -> Tuple[np.ndarray, np.ndarray]:
    rgba = np.stack(self.images).astype(np.uint8)
    flags = np.zeros(len(self.names), np.int32)
    for i, im in enumerate(self.images):
      a = im[..., 3]
      if ((a > 0) & (a < 255)).any():
        flags[i] |= SPRITE_FLAG_PARTIAL
      if (a == 255).all():
        flags[i] |= SPRITE_FLAG_OPAQUE
      if (a == 0).all():
        flags[i] |= SPRITE_FLAG_EMPTY
    return rgba, flags


def _parse_map(ascii_map: str) -> List[str]:
  # prefab_utils.lua:113-131 (_visitText): strip leading newlines only.
  text = ascii_map.lstrip("\n")
  rows = text.split("\n")
  while rows and rows[-1] == "":
    rows.pop()
  return rows


def _expand(spec, prefabs, choice=None) -> List[str]:
  """_createPrefabsFromSpec (prefab_utils.lua:44-72): the prefab names one map
  character creates, in creation order.  A plain name is itself (it must be a
  key of `prefabs`); {'type': 'all'} expands every element in list order;
  {'type': 'choice'} expands `choice(list)` — the reference's
  `random:choice(prefab.list)`.  Without a `choice` function a 'choice' spec is
  refused (callers enumerate its outcomes with `_alternatives`)."""
  if isinstance(spec, str):
    assert spec in prefabs, f"Prefab with name '{spec}' not found prefabs."
    return [spec]
  assert "type" in spec and "list" in spec, "a prefab spec is {type=..., list={...}}"
  if spec["type"] == "all":
    out = []
    for p in spec["list"]:
      out.extend(_expand(p, prefabs, choice))
    return out
  if spec["type"] == "choice":
    if choice is None:
      raise NotImplementedError("a 'choice' prefab spec nested inside another spec")
    return _expand(choice(list(spec["list"])), prefabs, choice)
  raise ValueError(f"unknown prefab spec type {spec['type']!r}")


def build_game_object_configs(ascii_map: str, prefabs, char_prefab_map,
                              choice=None) -> List[Tuple[str, int, int]]:
  """prefab_utils.buildGameObjectConfigs (prefab_utils.lua:163-177): the map's
  game objects in creation order, as (prefab name, x, y) — x = column, y = row
  (`transform.kwargs.position = {col, row}`, :39), orientation always 'N' (:40).
  Rows top to bottom, columns left to right (_visitText, :113-131); characters
  that are not in `char_prefab_map` are ignored (_processChar, :97-108).
  `choice(list)` stands for `random:choice(list)`; the lowering never passes it
  (it enumerates the outcomes of a 'choice' character instead, `_alternatives`),
  the KAT tests pass the reference tests' mock."""
  return [(pname, x, y) for x, y, spec in _visit_map(ascii_map, char_prefab_map)
          for pname in _expand(spec, prefabs, choice)]


def _visit_map(ascii_map: str, char_prefab_map):
  """(x, y, prefab spec) of every map character that has one, in the order the
  reference visits them (_visitText + _processChar, prefab_utils.lua:97-131)."""
  for y, row in enumerate(_parse_map(ascii_map)):
    for x, ch in enumerate(row):
      spec = char_prefab_map.get(ch)
      if spec is not None:
        yield x, y, spec


def _alternatives(spec, prefabs) -> List[List[str]]:
  """The outcomes of one map character (prefab_utils.lua:94-109).  A 'choice'
  spec is `random:choice(list)` once per world build, i.e. per episode: one
  alternative per list entry, repeats included (they carry the odds); anything
  else has a single outcome."""
  if isinstance(spec, dict) and spec["type"] == "choice":
    return [_expand(el, prefabs) for el in spec["list"]]
  return [_expand(spec, prefabs)]


def _kind_of(obj) -> int:
  names = {c["component"] for c in obj["components"]}
  if "Avatar" in names:
    return KIND_AVATAR
  if "AppleGrow" in names:
    return KIND_APPLE_GROW
  if "DirtTracker" in names:
    return KIND_DIRT
  if "DensityRegrow" in names:
    return KIND_DENSITY_REGROW
  if "Resource" in names:
    return KIND_RESOURCE
  if "RewardIndicator" in names:
    return KIND_REWARD_INDICATOR
  if "GraduatedSanctionsMarking" in names:
    return KIND_MARKING
  if "ReadyToInteractMarker" in names:
    return KIND_READY_MARKER
  if "Coin" in names:
    return KIND_COIN
  if "Ore" in names:
    return KIND_ORE
  if "Pickable" in names:
    return KIND_TOKEN
  if "MushroomEating" in names:
    return KIND_MUSHROOM
  if "Container" in names:
    return KIND_CONTAINER
  if "Receiver" in names:
    return KIND_RECEIVER
  if "CookingPot" in names:
    return KIND_POT
  if "Inventory" in names and "playerIndex" in (_get_component(obj, "Inventory").get("kwargs") or {}):
    return KIND_INVENTORY
  if "LoadingBarVisualiser" in names:
    return KIND_LOADING_BAR
  if obj.get("name") == "resource_texture":
    return KIND_TEXTURE
  if obj.get("name") == "damage_indicator":
    return KIND_DAMAGE_INDICATOR
  if "Animation" in names:
    return KIND_ANIM
  return KIND_STATIC


def _names_blob(names: Sequence[str]) -> np.ndarray:
  return np.frombuffer(("\0".join(names) + "\0").encode(), np.uint8).copy()


def lower_common(settings: Mapping[str, Any],
                 extra_layers: Sequence[str] = ()) -> Dict[str, Any]:
  """Substrate-independent part of the lowering.  Returns a dict with numpy
  tables plus python-side helper structures under keys starting with '_'.
  `extra_layers`: layers a level's Simulation subclass appends to renderOrder
  (territory/init.lua:30-37)."""
  sim = settings["simulation"]
  size = int(settings.get("spriteSize", 16))
  rows = _parse_map(sim["map"])
  # (coins pads its procedurally generated map unevenly: the grid is as wide as
  # the longest row, shorter rows end in empty cells)
  H, W = len(rows), max(len(r) for r in rows)
  rows = [r + " " * (W - len(r)) for r in rows]
  prefabs = sim["prefabs"]
  cpm = sim["charPrefabMap"]
  # gameObjects: the avatars and the objects that go with them (territory's
  # markings after the avatars, the_matrix's readiness markers interleaved with
  # them); created in list order (base_simulation.lua:103-118)
  game_objects = list(sim["gameObjects"])
  avatars = [o for o in game_objects if _get_component(o, "Avatar")]
  P = int(settings["numPlayers"])
  assert len(avatars) >= P

  # ---- object list in creation order (base_simulation.lua:103-131)
  objects: List[Tuple[Mapping[str, Any], int, int]] = []
  obj_choice: List[Tuple[int, int]] = []   # per object: (choice id or -1, outcome mask)
  choice_n: List[int] = []                 # per choice: number of outcomes
  objects.append((sim["scene"], 0, 0))
  obj_choice.append((-1, 0))
  for av in game_objects:
    # (an explicit object may name its cell: collaborative_cooking.py:727-754 puts an
    # inventory over every container and a loading bar over every pot this way)
    pos = (_get_component(av, "Transform").get("kwargs") or {}).get("position") or (0, 0)
    objects.append((av, int(pos[0]), int(pos[1])))
    obj_choice.append((-1, 0))
  alt_maps = sim.get("mapAlternatives")
  if alt_maps:
    # ONE choice for the whole map: the config draws it in build() — coins.py:45-82
    # get_ascii_map: random width and height, padded to the largest size — so every
    # environment built has its own map and keeps it for all its episodes
    # (mapChoiceScope 'world').  The pack holds the union of the alternatives'
    # objects in row-major creation order, each with the set of maps it is part of.
    assert len(alt_maps) <= 64 and sim.get("mapChoiceScope") in ("world", "episode")
    cid = len(choice_n)
    choice_n.append(-len(alt_maps) if sim["mapChoiceScope"] == "world" else len(alt_maps))
    per_cell: Dict[Tuple[int, int], List[List[Any]]] = {}
    for k, amap in enumerate(alt_maps):
      assert [len(r) for r in _parse_map(amap)] and len(_parse_map(amap)) == H
      for x, y, spec in _visit_map(amap, cpm):
        alts = _alternatives(spec, prefabs)
        assert len(alts) == 1
        for pname in alts[0]:
          entries = per_cell.setdefault((y, x), [])
          for ent in entries:
            if ent[0] == pname:
              ent[1] |= 1 << k
              break
          else:
            entries.append([pname, 1 << k])
    full = (1 << len(alt_maps)) - 1
    for (y, x) in sorted(per_cell):
      for pname, mask in per_cell[(y, x)]:
        objects.append((prefabs[pname], x, y))
        obj_choice.append((-1, 0) if mask == full else (cid, mask))
  for x, y, spec in ([] if alt_maps else _visit_map(sim["map"], cpm)):
    if True:
      alts = _alternatives(spec, prefabs)
      if len(alts) == 1:
        for pname in alts[0]:
          objects.append((prefabs[pname], x, y))
          obj_choice.append((-1, 0))
        continue
      # per-episode choice: the pack holds the union of the alternatives'
      # objects, each with the set of outcomes it exists in (outcome k of choice
      # c = Philox draw RS_MAP_CHOICE, index c, bounded by the list length)
      assert len(alts) <= 31 and len(choice_n) < 65536
      cid = len(choice_n)
      choice_n.append(len(alts))
      seen: List[str] = []
      for alt in alts:
        assert len(set(alt)) == len(alt)
        for pname in alt:
          if pname not in seen:
            seen.append(pname)
      for pname in seen:
        mask = sum(1 << k for k, alt in enumerate(alts) if pname in alt)
        objects.append((prefabs[pname], x, y))
        obj_choice.append((-1, 0) if mask == (1 << len(alts)) - 1 else (cid, mask))

  # ---- layers: base render order + hit layers in registration order.
  # base_simulation.lua:281-284 iterates objects with addHits in creation order
  # and game_object.lua:215-222 iterates components; Lua's pairs() makes the
  # component order unspecified -> we fix config order.
  layers = list(BASE_RENDER_ORDER)
  hits: List[Tuple[str, str, str]] = []  # (hitName, layer, sprite)
  # layers that hold pieces: a hit whose sprite is drawn on such a layer gets a
  # virtual layer right above it (assumption A13: the beam sprite is composited
  # over the layer's own piece, it does not replace it)
  piece_layers = set()
  for obj, _, _ in objects:
    for cfg in _get_component(obj, "StateManager")["kwargs"]["stateConfigs"]:
      if isinstance(cfg.get("layer"), str):
        piece_layers.add(cfg["layer"])
  virtual_above: Dict[str, str] = {}

  def add_hit(hit, layer, sprite, render=True):
    if layer in piece_layers:
      virtual_above[layer] = layer + "#hits"
      layer = layer + "#hits"
    if hit not in [h[0] for h in hits]:
      hits.append((hit, layer, sprite))
    if render and layer not in layers and not layer.endswith("#hits"):
      layers.append(layer)

  sprites = _Sprites(size)
  # base_simulation.lua:322-324
  sprites.add_color("OutOfBounds", (0, 0, 0))
  sprites.add_color("OutOfView", (80, 80, 80))

  seen_prefab_sprites = set()
  for obj, _, _ in objects:
    for c in obj["components"]:
      kw = c.get("kwargs", {}) or {}
      name = c["component"]
      if name == "Zapper":
        # avatar_library.lua:597-607
        add_hit("zapHit", "beamZap", "BeamZap")
        sprites.add_color("BeamZap", kw.get("beamColor", (252, 252, 106)))
      elif name == "GameInteractionZapper":
        # the_matrix/components.lua:384-394
        add_hit("gameInteraction", "beamInteraction", "BeamInteraction")
        sprites.add_color("BeamInteraction", kw.get("beamColor", (252, 252, 106)))
      elif name == "Cleaner":
        # clean_up/components.lua:185-195
        add_hit("cleanHit", "beamClean", "BeamClean")
        sprites.add_color("BeamClean", (99, 223, 242, 175))
      elif name == "MineBeam":
        # coop_mining/components.lua:177-188
        add_hit("mine", "beamMine", "beamMine")
        sprites.add_color("beamMine", (255, 202, 202))
      elif name == "InteractBeam":
        # collaborative_cooking/components.lua:52-77: hit, layer and sprite share one name per
        # avatar ('interact_' .. the avatar's unique state; here: its index)
        hn = f"interact_{int(_get_component(obj, 'Avatar')['kwargs']['index'])}"
        sprites.add_shape(hn, kw["shapes"][0], kw["palettes"][0], True)
        add_hit(hn, hn, hn)
      elif name == "GiftBeam":
        # gift_refinements/components.lua:118-129
        add_hit("gift", "beamGift", "beamGift")
        sprites.add_color("beamGift", (255, 202, 202))
      elif name == "Paintbrush":
        # territory/components.lua:362-399: oriented sprite brush<i>.{N,E,S,W}
        i = int(kw["playerIndex"])
        sprites.add_shape(f"brush{i}", list(kw["shape"]), kw["palette"], True)
        add_hit(f"directionHit{i}", "directionIndicatorLayer", f"brush{i}",
                render=False)
      elif name == "ResourceClaimer":
        # territory/components.lua:243-253
        i = int(kw["playerIndex"])
        sprites.add_color(f"claimBeamSprite_{i}", kw["color"])
        add_hit(f"claimBeam_{i}", "superDirectionIndicatorLayer",
                f"claimBeamSprite_{i}", render=False)
      elif name == "Appearance":
        key = id(c)
        if key not in seen_prefab_sprites:
          seen_prefab_sprites.add(key)
          sprites.add_from_appearance(kw)
      elif name == "AdditionalSprites":
        key = id(c)
        if key not in seen_prefab_sprites:
          seen_prefab_sprites.add(key)
          sprites.add_from_appearance(kw, custom=True)

  for lname in extra_layers:
    if lname not in layers:
      layers.append(lname)
  for base, virt in virtual_above.items():
    if base not in layers:
      layers.append(base)
    layers.insert(layers.index(base) + 1, virt)

  # ---- states.  The reference gives every game object its own unique states
  # (game_object.lua getUniqueState) and the Lua rules compare state NAMES, so
  # the engine's state id is type-level: (object name, state name, its config).
  # Two prefabs of one name share ids where their state configs agree
  # (clean_up's two DirtContainer prefabs) and get separate ids where they do
  # not (commons_harvest's two spawnPoint prefabs differ in `groups`).
  # `state_ids[(id(prefab), state)]` resolves a prefab's own state;
  # `state_ids[(name, state)]` the first registered state of that name.
  state_ids: Dict[Tuple[Any, ...], int] = {}
  state_layer: List[int] = [-1]
  state_sprite: List[int] = [-1]
  state_groups: List[int] = [0]
  state_contact: List[int] = [-1]
  state_names: List[str] = ["<empty>"]
  groups: List[str] = []
  contacts: List[str] = []

  def gid(name):
    if name not in groups:
      groups.append(name)
    return groups.index(name)

  def ensure_states(obj):
    sm = _get_component(obj, "StateManager")
    oname = obj["name"]
    for cfg in sm["kwargs"]["stateConfigs"]:
      key = (oname, cfg["state"], cfg.get("layer"), cfg.get("sprite"),
             tuple(cfg.get("groups", []) or []), cfg.get("contact"))
      if key in state_ids:
        state_ids[(id(obj), cfg["state"])] = state_ids[key]
        continue
      state_ids[key] = len(state_layer)
      state_ids[(id(obj), cfg["state"])] = len(state_layer)
      state_ids.setdefault((oname, cfg["state"]), len(state_layer))
      layer = cfg.get("layer")
      if isinstance(layer, str) and layer not in layers:
        layers.append(layer)
      state_layer.append(layers.index(layer) if isinstance(layer, str) else -1)
      spr = cfg.get("sprite")
      state_sprite.append(sprites.index(spr) if isinstance(spr, str) else -1)
      mask = 0
      for g in cfg.get("groups", []) or []:
        mask |= 1 << gid(g)
      state_groups.append(mask)
      ct = cfg.get("contact")
      if isinstance(ct, str):
        if ct not in contacts:
          contacts.append(ct)
        state_contact.append(contacts.index(ct))
      else:
        state_contact.append(-1)
      state_names.append(f"{oname}.{cfg['state']}")

  for obj, _, _ in objects:
    ensure_states(obj)

  # pseudo-states for beam sprites living on the hit layer: one per hit, or one
  # per beam direction when the sprite has distinct facings (the beam sprite of
  # an oriented sprite faces the way the beam travels).
  hit_state = []
  hit_state_dir = []
  state_orient = [0] * len(state_layer)
  for hit, layer, sprite in hits:
    img = sprites.images[sprites.index(sprite)]
    oriented = any(not np.array_equal(img[0], img[d]) for d in range(1, 4))
    ids = []
    for d in range(4 if oriented else 1):
      ids.append(len(state_layer))
      state_layer.append(layers.index(layer))
      state_sprite.append(sprites.index(sprite))
      state_groups.append(0)
      state_contact.append(-1)
      state_orient.append(d)
      state_names.append(f"<hit>.{hit}" + (f".{COMPASS[d]}" if oriented else ""))
    state_ids[("<hit>", hit)] = ids[0]
    hit_state.append(ids[0])
    hit_state_dir.append([ids[d if oriented else 0] for d in range(4)])

  # BeamBlocker components (component_library.lua:667-685): every state of an
  # object carrying BeamBlocker{beamType=h} stops beams of hit h.
  state_hit_block = [0] * len(state_layer)
  hit_names = [h[0] for h in hits]
  for obj, _, _ in objects:
    blockers = [c for c in obj["components"] if c["component"] == "BeamBlocker"]
    block_all = any(c["component"] == "AllBeamBlocker" for c in obj["components"])
    if not blockers and not block_all:
      continue
    sm = _get_component(obj, "StateManager")
    for cfg in sm["kwargs"]["stateConfigs"]:
      sidx = state_ids[(id(obj), cfg["state"])]
      if block_all:  # territory/components.lua:37-49
        state_hit_block[sidx] = (1 << len(hit_names)) - 1
      for b in blockers:
        bt = b["kwargs"]["beamType"]
        if bt in hit_names:
          state_hit_block[sidx] |= 1 << hit_names.index(bt)

  assert len(state_layer) <= 255, "state ids are stored as u8"
  L = len(layers)

  # ---- object table + initial grid
  obj_tab = np.zeros((len(objects), 4), np.int32)
  init_grid = np.zeros((L, H, W), np.uint8)
  for i, (obj, x, y) in enumerate(objects):
    sm = _get_component(obj, "StateManager")["kwargs"]
    s0 = state_ids[(id(obj), sm["initialState"])]
    kind = KIND_SCENE if i == 0 else _kind_of(obj)
    obj_tab[i] = (kind, x, y, s0)
    if kind not in (KIND_SCENE, KIND_AVATAR):
      ly = state_layer[s0]
      if ly >= 0:
        if obj_choice[i][0] >= 0 and init_grid[ly, y, x] != 0:
          # alternatives of one 'choice' character on one layer (the_matrix's
          # resource classes): at most one exists in an episode; the reset writes
          # the state of the one that does (optional_i32)
          continue
        assert init_grid[ly, y, x] == 0, (
            f"two pieces on layer {layers[ly]} at {(x, y)}")
        init_grid[ly, y, x] = s0

  # ---- avatars
  alive, wait = [], []
  view = None
  sprite_map = np.tile(np.arange(len(sprites.names), dtype=np.int32),
                       (P + 1, 1))
  action_names = None
  action_spec = None
  for p in range(P):
    av = avatars[p]
    akw = _get_component(av, "Avatar")["kwargs"]
    assert int(akw["index"]) == p + 1
    alive.append(state_ids[(id(av), akw["aliveState"])])
    wait.append(state_ids[(id(av), akw["waitState"])])
    v = akw["view"]
    vv = (int(v["left"]), int(v["right"]), int(v["forward"]),
          int(v["backward"]))
    assert not v.get("centered", False)
    assert view in (None, vv)
    view = vv
    for src, dst in (akw.get("spriteMap") or {}).items():
      sprite_map[p, sprites.index(src)] = sprites.index(dst)
    order = tuple(akw.get("actionOrder", ("move", "turn")))
    assert action_names in (None, order)
    action_names = order
    # the raw action fields dmlab2d exposes as "<player>.<name>" (avatar_library.lua:
    # 205-223 Avatar:discreteActionSpec / discreteActions): (min, max, default) per field, in actionOrder
    # (Avatar.__init__'s default, avatar_library.lua:67-72, for a config that names none)
    aspec = akw.get("actionSpec") or {"move": {"default": 0, "min": 0, "max": 4},
                                      "turn": {"default": 0, "min": -1, "max": 1}}
    spec = tuple((int(aspec[n]["min"]), int(aspec[n]["max"]),
                  int(aspec[n].get("default", 0))) for n in order)
    assert action_spec in (None, spec) and len(order) <= 4
    assert all(-128 <= lo <= d <= hi <= 127 for lo, hi, d in spec)
    # (the engine packs a fourth field into six unsigned bits: mp_engine.hip)
    assert len(spec) < 4 or 0 <= spec[3][0] <= spec[3][1] <= 63
    action_spec = spec
  world_map = sim.get("worldSpriteMap") or {}
  for src, dst in world_map.items():
    sprite_map[P, sprites.index(src)] = sprites.index(dst)

  avatar_layer = state_layer[alive[0]]
  assert all(state_layer[s] == avatar_layer for s in alive)

  rgba, sflags = sprites.arrays()
  hdr = np.zeros(HDR_LEN, np.int32)
  hdr[HDR_VERSION] = 1
  hdr[HDR_H], hdr[HDR_W], hdr[HDR_L] = H, W, L
  hdr[HDR_NSTATES] = len(state_layer)
  hdr[HDR_NSPRITES] = len(sprites.names)
  hdr[HDR_P] = P
  hdr[HDR_SPRITE] = size
  hdr[HDR_TOPOLOGY] = {"BOUNDED": 0, "TORUS": 1}[settings.get("topology",
                                                              "BOUNDED")]
  hdr[HDR_VL:HDR_VB + 1] = view
  hdr[HDR_MAXFRAMES] = int(settings.get("maxEpisodeLengthFrames", 3600))
  hdr[HDR_NOBJ] = len(objects)
  hdr[HDR_NGROUPS] = len(groups)
  hdr[HDR_AVATAR_LAYER] = avatar_layer
  hdr[HDR_NHITS] = len(hits)
  hdr[HDR_NFIELDS] = len(action_names)

  spawn_mask = 1 << groups.index("spawnPoints") if "spawnPoints" in groups else 0

  # ---- spawn groups (avatar_library.lua:108-124,322-327; base_simulation.lua:
  # 396-445): the first spawn uses `spawnGroup`, later ones
  # `postInitialSpawnGroup` (default: the same).  Cells of a group in piece
  # creation order; initial groups in order of first use by player index.
  def cells_of_group(gname):
    m = 1 << groups.index(gname)
    return [y * W + x for (o, x, y), row in zip(objects, obj_tab)
            if row[0] not in (KIND_SCENE, KIND_AVATAR) and state_groups[row[3]] & m]
  init_groups: List[str] = []
  avatar_init_group = []
  respawn_groups = set()
  for p in range(P):
    akw = _get_component(avatars[p], "Avatar")["kwargs"]
    g0 = akw["spawnGroup"]
    g1 = akw.get("postInitialSpawnGroup", "_DEFAULT")
    respawn_groups.add(g0 if g1 == "_DEFAULT" else g1)
    if g0 not in init_groups:
      init_groups.append(g0)
    avatar_init_group.append(init_groups.index(g0))
  assert len(respawn_groups) == 1, "per-avatar respawn groups are not lowered"
  init_cells, init_ptr = [], [0]
  for g in init_groups:
    init_cells += cells_of_group(g)
    init_ptr.append(len(init_cells))
  respawn_cells = cells_of_group(next(iter(respawn_groups)))

  out = {
      "hdr": hdr,
      "layer_names": _names_blob(layers),
      "state_names": _names_blob(state_names),
      "sprite_names": _names_blob(sprites.names),
      "group_names": _names_blob(groups),
      "state_layer": np.asarray(state_layer, np.int32),
      "state_sprite": np.asarray(state_sprite, np.int32),
      "state_groups": np.asarray(state_groups, np.uint32),
      "state_contact": np.asarray(state_contact, np.int32),
      "sprite_rgba": rgba,
      "sprite_flags": sflags,
      "init_grid": init_grid,
      "objects": obj_tab,
      "avatar_alive_state": np.asarray(alive, np.int32),
      "avatar_wait_state": np.asarray(wait, np.int32),
      "view_sprite_map": sprite_map,
      "hit_state": np.asarray(hit_state, np.int32),
      "hit_state_dir": np.asarray(hit_state_dir, np.int32),
      "state_orient": np.asarray(state_orient, np.int32),
      "state_hit_block": np.asarray(state_hit_block, np.uint32),
      "hit_names": _names_blob(hit_names),
      "spawn_cells": np.asarray(respawn_cells, np.int32),
      "init_spawn_cells": np.asarray(init_cells, np.int32),
      "init_spawn_ptr": np.asarray(init_ptr, np.int32),
      "avatar_init_group": np.asarray(avatar_init_group, np.int32),
      # group bit (in state_groups) of each initial spawn group: a 'choice' spawn
      # point is present in an episode iff a piece of that group stands on its cell
      "init_spawn_mask": np.asarray([1 << groups.index(g) for g in init_groups], np.uint32),
      "action_spec": np.asarray(action_spec, np.int32).reshape(-1, 3),
      "action_names": np.frombuffer(b"".join(n.encode() + b"\0" for n in action_names), np.uint8),
      "_layers": layers,
      "_groups": groups,
      "_state_ids": state_ids,
      "_sprites": sprites.names,
      "_objects": objects,
      "_avatars": avatars[:P],
      "_action_names": action_names,
      "_hits": hits,
      "_spawn_mask": spawn_mask,
  }
  if choice_n:
    # per-episode 'choice' prefabs: outcomes per choice, (choice, outcome mask) per
    # object, and for the engine the grid cells to clear when an object is absent
    # choice_n[c] < 0: -choice_n[c] outcomes drawn once per WORLD (not per episode);
    # masks of choices with more than 32 outcomes continue in object_choice_hi, and
    # in the engine's rows as a second row whose choice word carries the first
    # outcome it covers (choice | 32 << 16)
    out["choice_n"] = np.asarray(choice_n, np.int32)
    lo32 = lambda m: int(np.int32(np.uint32(m & 0xffffffff)))
    out["object_choice"] = np.asarray([(c, lo32(m)) for c, m in obj_choice], np.int32).reshape(-1, 2)
    if any(m >> 32 for _, m in obj_choice):
      out["object_choice_hi"] = np.asarray([lo32(m >> 32) for _, m in obj_choice], np.int32)
    opt = []
    for i, ((obj, x, y), (cid, mask)) in enumerate(zip(objects, obj_choice)):
      if cid < 0:
        continue
      s0 = int(obj_tab[i, 3])
      ly = state_layer[s0]
      if ly >= 0:
        # plane | initial state << 8: the reset clears the cell of every optional
        # object, then writes the state of those that exist this episode
        opt.append((y * W + x, ly | (s0 << 8), cid, lo32(mask)))
        if mask >> 32:
          opt.append((y * W + x, ly | (s0 << 8), cid | (32 << 16), lo32(mask >> 32)))
    out["optional_i32"] = np.asarray(opt, np.int32).reshape(-1, 4)
  return out


def _cells_of_kind(obj_tab: np.ndarray, kind: int, W: int) -> np.ndarray:
  sel = obj_tab[obj_tab[:, 0] == kind]
  return (sel[:, 2] * W + sel[:, 1]).astype(np.int32)


def _zapper_tables(av) -> Tuple[np.ndarray, np.ndarray]:
  kw = _get_component(av, "Zapper")["kwargs"]
  zi = np.asarray([
      int(kw["cooldownTime"]), int(kw["beamLength"]), int(kw["beamRadius"]),
      int(kw["framesTillRespawn"]), int(bool(kw.get("removeHitPlayer", True)))
  ], np.int32)
  zf = np.asarray([float(kw["penaltyForBeingZapped"]),
                   float(kw["rewardForZapping"])], np.float64)
  return zi, zf


def _action_table(action_set, names) -> np.ndarray:
  tab = np.zeros((len(action_set), 4), np.int32)
  for i, row in enumerate(action_set):
    for j, n in enumerate(names):
      tab[i, j] = int(row.get(n, 0))
  return tab


def clean_up_apple_thresholds(n: int, max_rate: float, dep: float,
                              rest: float) -> np.ndarray:
  """AppleGrow:update (clean_up/components.lua:64-80) as a function of the dirt
  count only (dirt + clean == number of dirt containers, always): the integer
  threshold of `uniformReal(0, 1) < probability` for each dirt count."""
  thr = np.zeros(n + 1, np.uint64)
  for d in range(n + 1):
    dirt_fraction = d / (d + (n - d))
    interp = (dirt_fraction - dep) / (rest - dep)
    interp = min(interp, 1.0)
    thr[d] = prob_threshold(float(max_rate) * interp)
  return thr


def lower_clean_up(settings: Mapping[str, Any], action_set) -> Dict[str, np.ndarray]:
  """clean_up: reference `configs/substrates/clean_up.py`,
  `lua/levels/clean_up/components.lua`."""
  t = lower_common(settings)
  hdr = t["hdr"]
  hdr[HDR_SUBSTRATE] = SUBSTRATE_IDS["clean_up"]
  W = int(hdr[HDR_W])
  sid = t["_state_ids"]
  objs = t["objects"]
  assert t["_action_names"] == ("move", "turn", "fireZap", "fireClean")
  t["action_table"] = _action_table(action_set, t["_action_names"])
  hdr[HDR_NACT] = len(action_set)

  av0 = t["_avatars"][0]
  t["zapper_i32"], t["zapper_f64"] = _zapper_tables(av0)
  ck = _get_component(av0, "Cleaner")["kwargs"]
  taste = _get_component(av0, "Taste")["kwargs"]
  assert taste.get("role", "free") == "free"

  scene = settings["simulation"]["scene"]
  ds = _get_component(scene, "DirtSpawner")["kwargs"]
  ee = _get_component(scene, "StochasticIntervalEpisodeEnding")["kwargs"]
  prefabs = settings["simulation"]["prefabs"]
  apple = prefabs["potential_apple"]
  ag = _get_component(apple, "AppleGrow")["kwargs"]
  ed = _get_component(apple, "Edible")["kwargs"]
  water = prefabs["river"]
  an = _get_component(water, "Animation")["kwargs"]

  t["apple_cells"] = _cells_of_kind(objs, KIND_APPLE_GROW, W)
  t["dirt_cells"] = _cells_of_kind(objs, KIND_DIRT, W)
  t["water_cells"] = _cells_of_kind(objs, KIND_ANIM, W)

  dirt = prefabs["potential_dirt"]
  t["cu_states"] = np.asarray(
      [sid[(id(apple), ed["liveState"])],
       sid[(id(apple), ed["waitState"])],
       sid[(id(dirt), "dirt")], sid[(id(dirt), "dirtWait")]] +
      [sid[(id(water), s)] for s in an["states"]], np.int32)
  assert len(an["states"]) == 4 and an["loop"] and an["randomStartFrame"]

  t["cu_i32"] = np.asarray([
      int(ck["cooldownTime"]), int(ck["beamLength"]), int(ck["beamRadius"]),
      int(ds["delayStartOfDirtSpawning"]),
      int(ee["minimumFramesPerEpisode"]), int(ee["intervalLength"]),
      int(an["gameFramesPerAnimationFrame"]),
  ], np.int32)
  t["cu_f64"] = np.asarray([
      float(ag["maxAppleGrowthRate"]), float(ag["thresholdDepletion"]),
      float(ag["thresholdRestoration"]), float(ds["dirtSpawnProbability"]),
      float(ee["probabilityTerminationPerInterval"]),
      float(ed["rewardForEating"]),
  ], np.float64)

  t["apple_thr"] = clean_up_apple_thresholds(len(t["dirt_cells"]),
                                             *[float(v) for v in t["cu_f64"][0:3]])
  t["thr_misc"] = np.asarray([prob_threshold(float(t["cu_f64"][3])),
                              prob_threshold(float(t["cu_f64"][4]))], np.uint64)
  return {k: v for k, v in t.items() if not k.startswith("_")}


def lower_commons_harvest(settings: Mapping[str, Any], action_set) -> Dict[str, np.ndarray]:
  """commons_harvest: reference `configs/substrates/commons_harvest__open.py`,
  `lua/levels/commons_harvest/components.lua` (DensityRegrow, Neighborhoods),
  `lua/modules/component_library.lua:953-1004` (Edible)."""
  t = lower_common(settings)
  hdr = t["hdr"]
  hdr[HDR_SUBSTRATE] = SUBSTRATE_IDS["commons_harvest"]
  W = int(hdr[HDR_W])
  sid = t["_state_ids"]
  objs = t["objects"]
  assert t["_action_names"] == ("move", "turn", "fireZap")
  t["action_table"] = _action_table(action_set, t["_action_names"])
  hdr[HDR_NACT] = len(action_set)
  t["zapper_i32"], t["zapper_f64"] = _zapper_tables(t["_avatars"][0])

  prefabs = settings["simulation"]["prefabs"]
  apple, grass = prefabs["apple"], prefabs["grass"]
  ed = _get_component(apple, "Edible")["kwargs"]
  dr = _get_component(apple, "DensityRegrow")["kwargs"]
  ee = _get_component(settings["simulation"]["scene"],
                      "StochasticIntervalEpisodeEnding")["kwargs"]
  assert ed["liveState"] == dr["liveState"] and ed["waitState"] == dr["waitState"]
  radius = float(dr["radius"])
  # DensityRegrow.__init__ (components.lua:93-98)
  nk = int(math.floor(math.pi * radius ** 2 + 1) + 1) if radius >= 0 else 0
  probs = [float(x) for x in dr["regrowthProbabilities"]]
  t["apple_cells"] = _cells_of_kind(objs, KIND_DENSITY_REGROW, W)
  t["ch_states"] = np.asarray(
      [sid[(id(apple), ed["liveState"])], sid[(id(apple), ed["waitState"])],
       sid[(id(grass), "grass")], sid[(id(grass), "dessicated")]] +
      [sid[(id(apple), f"{dr['waitState']}_{k}")] for k in range(nk)], np.int32)
  t["ch_i32"] = np.asarray([
      nk, int(ee["minimumFramesPerEpisode"]), int(ee["intervalLength"]),
      int(bool(dr.get("canRegrowIfOccupied", True))),
  ], np.int32)
  t["ch_f64"] = np.asarray([float(ed["rewardForEating"]), radius,
                            float(ee["probabilityTerminationPerInterval"])] + probs,
                           np.float64)
  # regrowth probability of wait group k: regrowthProbabilities[min(k, n - 1)]
  # (components.lua:119-136), as exact integer thresholds; last = episode end
  t["ch_thr"] = np.asarray(
      [prob_threshold(probs[min(k, len(probs) - 1)]) for k in range(nk)] +
      [prob_threshold(float(ee["probabilityTerminationPerInterval"]))], np.uint64)
  # queryDisc(layer, radius): cells with dx^2 + dy^2 <= radius^2, self excluded
  r = int(math.floor(radius))
  disc = [(dx, dy) for dy in range(-r, r + 1) for dx in range(-r, r + 1)
          if (dx or dy) and dx * dx + dy * dy <= radius * radius]
  t["disc_offsets"] = np.asarray(disc, np.int32).reshape(-1, 2)
  return {k: v for k, v in t.items() if not k.startswith("_")}


# Components each lowering understands.  Anything else carries rules this
# engine does not implement: refuse rather than silently drop them.
_COMMON_COMPONENTS = {
    "StateManager", "Transform", "Appearance", "AdditionalSprites", "BeamBlocker",
    "Avatar", "Zapper", "ReadyToShootObservation", "StochasticIntervalEpisodeEnding",
}
_LEVEL_COMPONENTS = {
    "clean_up": {"AllNonselfCumulants", "Animation", "AppleGrow",
                 "AvatarMetricReporter", "Cleaner", "DirtCleaning", "DirtSpawner",
                 "DirtTracker", "Edible", "GlobalData", "RiverMonitor", "Taste"},
    "commons_harvest": {"DensityRegrow", "Edible", "Neighborhoods"},
    "territory": {"AllBeamBlocker", "Resource", "RewardIndicator", "Paintbrush",
                  "ResourceClaimer", "Taste", "GraduatedSanctionsMarking"},
    "coins": {"Coin", "ChoiceCoinRegrow", "PlayerCoinType", "Role", "PartnerTracker",
              "GlobalCoinCollectionTracker", "GlobalMetricReporter",
              "AvatarMetricReporter"},
    "coop_mining": {"Ore", "FixedRateRegrow", "MineBeam", "MiningTracker"},
    "gift_refinements": {"FixedRateRegrow", "Pickable", "GiftBeam", "Inventory", "TokenTracker",
                         "AvatarMetricReporter"},
    "collaborative_cooking": {"InteractBeam", "Container", "Inventory", "Receiver", "CookingPot",
                              "LoadingBarVisualiser", "AvatarCumulants"},
    "externality_mushrooms": {"MushroomEating", "MushroomGrowable", "MushroomRegrowth", "Destroyable",
                              "Perishable", "Cumulants", "GraduatedSanctionsMarking"},
    "the_matrix": {"TheMatrix", "Resource", "Destroyable", "GameInteractionZapper",
                   "InventoryObserver", "SpawnResourcesWhenAllPlayersZapped", "Taste",
                   "InteractionTaste", "DyadicRole", "AvatarMetricReporter",
                   "AvatarConnector", "ReadyToInteractMarker"},
}


def check_components(settings: Mapping[str, Any]) -> None:
  level = settings["levelName"]
  known = _COMMON_COMPONENTS | _LEVEL_COMPONENTS.get(level, set())
  sim = settings["simulation"]
  objs = ([sim["scene"]] if "scene" in sim else []) + list(sim["gameObjects"]) + list(sim["prefabs"].values())
  unknown = {c["component"] for o in objs for c in o["components"]} - known
  # Role + RoleBasedRewardTile (avatar_library.lua:1178-1203,
  # component_library.lua:1097-1133): a tile that rewards the avatars whose Role
  # is in its table when they step on it.  With no such avatar it never fires
  # (commons_harvest__partnership: every role is "none", the table is
  # {"putative_cooperator": -10}) and both components are inert.
  if unknown & {"Role", "RoleBasedRewardTile"}:
    roles = {c["kwargs"]["role"] for o in objs for c in o["components"]
             if c["component"] == "Role"}
    rewarded = set()
    for o in objs:
      for c in o["components"]:
        if c["component"] == "RoleBasedRewardTile":
          rewarded |= set(c["kwargs"]["rolesToRewards"])
    if not (roles & rewarded):
      unknown -= {"Role", "RoleBasedRewardTile"}
  unknown = sorted(unknown)
  if unknown:
    raise NotImplementedError(
        f"level {level!r}: components {unknown} are not implemented by the engine")


def lower_territory(settings: Mapping[str, Any], action_set) -> Dict[str, np.ndarray]:
  """territory: reference `configs/substrates/territory.py` (+ `territory__rooms.py`
  for the map), `lua/levels/territory/{init,components}.lua`,
  `lua/modules/avatar_library.lua:948-1121` (GraduatedSanctionsMarking)."""
  t = lower_common(settings, extra_layers=("directionIndicatorLayer",
                                           "superDirectionIndicatorLayer"))
  hdr = t["hdr"]
  hdr[HDR_SUBSTRATE] = SUBSTRATE_IDS["territory"]
  W, P = int(hdr[HDR_W]), int(hdr[HDR_P])
  sid = t["_state_ids"]
  objs = t["objects"]
  assert t["_action_names"] == ("move", "turn", "fireZap", "fireClaim")
  t["action_table"] = _action_table(action_set, t["_action_names"])
  hdr[HDR_NACT] = len(action_set)
  av0 = t["_avatars"][0]
  t["zapper_i32"], t["zapper_f64"] = _zapper_tables(av0)
  t["zapper_i32"][3] = min(int(t["zapper_i32"][3]), 1 << 30)  # framesTillRespawn 1e6
  assert not t["zapper_i32"][4], "territory: GraduatedSanctionsMarking removes, not Zapper"

  prefabs = settings["simulation"]["prefabs"]
  res, tex = prefabs["resource"], prefabs["resource_texture"]
  ind, dmg = prefabs["reward_indicator"], prefabs["damage_indicator"]
  marking = [o for o in settings["simulation"]["gameObjects"]
             if _get_component(o, "GraduatedSanctionsMarking")]
  assert len(marking) == P
  rk = _get_component(res, "Resource")["kwargs"]
  ck = _get_component(av0, "ResourceClaimer")["kwargs"]
  gk = _get_component(marking[0], "GraduatedSanctionsMarking")["kwargs"]
  taste = _get_component(av0, "Taste")["kwargs"]
  assert taste.get("role", "none") == "none"
  ee = _get_component(settings["simulation"]["scene"],
                      "StochasticIntervalEpisodeEnding")["kwargs"]
  assert rk["destroyedState"] == "destroyed" and gk["hitName"] == "zapHit"
  assert int(gk.get("initialLevel", 1)) == 1
  logic = gk["hitLogic"]

  t["resource_cells"] = _cells_of_kind(objs, KIND_RESOURCE, W)
  t["tr_states"] = np.asarray(
      [sid[(id(res), "unclaimed")], sid[(id(res), "destroyed")],
       sid[(id(tex), "unclaimed")], sid[(id(tex), "destroyed")],
       sid[(id(ind), "inactive")],
       sid[(id(dmg), "inactive")], sid[(id(dmg), "damaged")],
       sid[(id(marking[0]), "level_1")], sid[(id(marking[0]), "level_2")],
       sid[(id(marking[0]), gk["waitState"])]] +
      [sid[(id(res), f"claimed_by_{i + 1}")] for i in range(P)] +
      [sid[(id(ind), f"dry_claimed_by_{i + 1}")] for i in range(P)], np.int32)
  for m in marking:  # one shared set of marking states
    assert sid[(id(m), "level_1")] == sid[(id(marking[0]), "level_1")]
  t["tr_i32"] = np.asarray(
      [int(rk["initialHealth"]), int(rk["rewardDelay"]),
       int(rk.get("delayTillSelfRepair", 15)), int(ck["beamLength"]),
       int(ck["beamRadius"]), int(ck["beamWait"]), int(gk["recoveryTime"]),
       len(logic), int(ee["minimumFramesPerEpisode"]), int(ee["intervalLength"])] +
      [v for lv in logic for v in (int(lv["levelIncrement"]), int(lv.get("freeze") or 0),
                                    int(bool(lv.get("remove", False))))], np.int32)
  t["tr_f64"] = np.asarray(
      [float(rk["reward"]), float(rk["rewardRate"]),
       float(rk.get("selfRepairProbability", 0.1)),
       float(ee["probabilityTerminationPerInterval"])] +
      [v for lv in logic for v in (float(lv["sourceReward"]), float(lv["targetReward"]))],
      np.float64)
  t["tr_thr"] = np.asarray([prob_threshold(float(rk["rewardRate"])),
                            prob_threshold(float(rk.get("selfRepairProbability", 0.1))),
                            prob_threshold(float(ee["probabilityTerminationPerInterval"]))],
                           np.uint64)
  # Stacks the renderer should pre-blend before any other triple (mp_create's
  # composite cache): a claimed resource that pays shows texture + wet paint + dry
  # paint of the SAME player (Resource / RewardIndicator, components.lua:60-200) —
  # 9 of the 81 combinations a cell's states allow, and most of the map late in
  # an episode.
  t["composite_hints"] = np.asarray(
      [[sid[(id(tex), "unclaimed")], sid[(id(res), f"claimed_by_{i + 1}")],
        sid[(id(ind), f"dry_claimed_by_{i + 1}")]] for i in range(P)], np.int32).reshape(-1)
  hit_names = [h[0] for h in t["_hits"]]
  t["tr_hits"] = np.asarray(
      [hit_names.index("zapHit")] +
      [hit_names.index(f"directionHit{i + 1}") for i in range(P)] +
      [hit_names.index(f"claimBeam_{i + 1}") for i in range(P)], np.int32)
  return {k: v for k, v in t.items() if not k.startswith("_")}


def lower_coins(settings: Mapping[str, Any], action_set) -> Dict[str, np.ndarray]:
  """coins: reference `configs/substrates/coins.py`, `lua/levels/coins/components.lua`
  (Coin, ChoiceCoinRegrow, PlayerCoinType, Role, PartnerTracker,
  GlobalCoinCollectionTracker).  The config draws the map size and the two coin
  colours with Python's `random` inside build(): a pack is ONE such instance
  (tools/make_packs.py seeds the generator)."""
  t = lower_common(settings)
  hdr = t["hdr"]
  hdr[HDR_SUBSTRATE] = SUBSTRATE_IDS["coins"]
  W, P = int(hdr[HDR_W]), int(hdr[HDR_P])
  sid = t["_state_ids"]
  objs = t["objects"]
  assert P == 2, "Coin:onEnter asserts at most 2 players (components.lua:95-96)"
  assert t["_action_names"] == ("move", "turn")
  t["action_table"] = _action_table(action_set, t["_action_names"])
  hdr[HDR_NACT] = len(action_set)
  coin = settings["simulation"]["prefabs"]["coin"]
  ck = _get_component(coin, "Coin")["kwargs"]
  rk = _get_component(coin, "ChoiceCoinRegrow")["kwargs"]
  ee = _get_component(settings["simulation"]["scene"],
                      "StochasticIntervalEpisodeEnding")["kwargs"]
  assert not ck.get("terminateEpisode", False), "coinsToTerminateEpisode is not lowered"
  assert rk["waitState"] == ck["waitState"]
  live = [rk["liveStateA"], rk["liveStateB"]]
  t["coin_cells"] = _cells_of_kind(objs, KIND_COIN, W)
  t["co_states"] = np.asarray([sid[(id(coin), live[0])], sid[(id(coin), live[1])],
                               sid[(id(coin), ck["waitState"])]], np.int32)
  # per player: index of its coin type among the live states (PlayerCoinType)
  ptype = []
  for av in t["_avatars"][:P]:
    ptype.append(live.index(_get_component(av, "PlayerCoinType")["kwargs"]["coinType"]))
  t["co_i32"] = np.asarray(ptype + [int(ee["minimumFramesPerEpisode"]),
                                    int(ee["intervalLength"])], np.int32)
  # per player: rewards it earns / the others earn for a match / a mismatch,
  # Role multipliers applied (components.lua:253-273)
  rew = []
  for av in t["_avatars"][:P]:
    role = _get_component(av, "Role")["kwargs"]
    rew += [float(ck["rewardSelfForMatch"]) * float(role.get("multiplyRewardSelfForMatch", 1.0)),
            float(ck["rewardSelfForMismatch"]) * float(role.get("multiplyRewardSelfForMismatch", 1.0)),
            float(ck["rewardOtherForMatch"]) * float(role.get("multiplyRewardOtherForMatch", 1.0)),
            float(ck["rewardOtherForMismatch"]) * float(role.get("multiplyRewardOtherForMismatch", 1.0))]
  t["co_f64"] = np.asarray(rew + [float(rk["regrowRate"]),
                                  float(ee["probabilityTerminationPerInterval"])], np.float64)
  t["co_thr"] = np.asarray([prob_threshold(float(rk["regrowRate"])),
                            prob_threshold(float(ee["probabilityTerminationPerInterval"]))],
                           np.uint64)
  colours = settings["simulation"].get("coinColours")
  if colours:
    # per-world colours: the coin's state per colour, each avatar's alive state per
    # colour, and (state, player) of those avatar states for the engine's
    # state -> player tables; the pair the instance itself drew, for reference
    assert len(colours) == 5
    t["co_colour_coin"] = np.asarray([sid[(id(coin), c)] for c in colours], np.int32)
    alive_c = [[sid[(id(av), f"{_get_component(av, 'Avatar')['kwargs']['aliveState']}_{c}")]
                for c in colours] for av in t["_avatars"][:P]]
    t["co_colour_alive"] = np.asarray(alive_c, np.int32)
    t["avatar_extra_alive"] = np.asarray(
        [(s, p) for p in range(P) for s in alive_c[p]], np.int32).reshape(-1, 2)
    t["co_colour_instance"] = np.asarray([colours.index(live[0]), colours.index(live[1])], np.int32)
  return {k: v for k, v in t.items() if not k.startswith("_")}


def lower_coop_mining(settings: Mapping[str, Any], action_set) -> Dict[str, np.ndarray]:
  """coop_mining: reference `configs/substrates/coop_mining.py`,
  `lua/levels/coop_mining/components.lua` (FixedRateRegrow :29-60, Ore :62-143,
  MineBeam :147-254, MiningTracker :256-283).  An ore object carries one `Ore`
  component per ore type over ONE state machine (oreWait / <type>Raw / <type>Partial);
  `FixedRateRegrow` grows type i = liveStates[i] with liveRates[i].  The Lua passes
  `minNumMiners` where the reward tables expect an ore-type index
  (Ore:onHit -> processRoleMineEvent(self._config.minNumMiners), :124,129): type k is
  the component with minNumMiners == k + 1, which the stock config satisfies (iron 1,
  gold 2) and this lowering requires."""
  t = lower_common(settings)
  hdr = t["hdr"]
  hdr[HDR_SUBSTRATE] = SUBSTRATE_IDS["coop_mining"]
  W, P = int(hdr[HDR_W]), int(hdr[HDR_P])
  sid = t["_state_ids"]
  objs = t["objects"]
  assert t["_action_names"] == ("move", "turn", "mine")
  assert P <= 8, "the miners of an ore are kept as a byte mask: at most 8 players"
  t["action_table"] = _action_table(action_set, t["_action_names"])
  hdr[HDR_NACT] = len(action_set)
  ore = settings["simulation"]["prefabs"]["ore"]
  ores = [c["kwargs"] for c in _components(ore, "Ore")]
  rk = _get_component(ore, "FixedRateRegrow")["kwargs"]
  ee = _get_component(settings["simulation"]["scene"],
                      "StochasticIntervalEpisodeEnding")["kwargs"]
  live = list(rk["liveStates"])
  K = len(live)
  assert K == 2 == len(ores) == len(rk["liveRates"]), "two ore types"
  assert all(o["waitState"] == rk["waitState"] for o in ores)
  # type k <-> the Ore component whose rawState is liveStates[k]; its minNumMiners is the
  # (1-based) ore-type index the Lua hands to the reward tables
  by_type = []
  for k, ls in enumerate(live):
    (o,) = [o for o in ores if o["rawState"] == ls]
    assert int(o["minNumMiners"]) == k + 1, "minNumMiners doubles as the ore-type index"
    assert int(o["miningWindow"]) >= 1
    by_type.append(o)
  assert by_type[0]["partialState"] == by_type[0]["rawState"], "a one-miner ore has no partial state"
  # (the order the components sit in the prefab is the order their onHit / update run:
  # no state is claimed by both, so the order is immaterial)
  t["ore_cells"] = _cells_of_kind(objs, KIND_ORE, W)
  t["cm_states"] = np.asarray(
      [sid[(id(ore), rk["waitState"])]] +
      [sid[(id(ore), o["rawState"])] for o in by_type] +
      [sid[(id(ore), o["partialState"])] for o in by_type], np.int32)
  av0 = t["_avatars"][0]
  mk = _get_component(av0, "MineBeam")["kwargs"]
  hit_names = [h[0] for h in t["_hits"]]
  t["cm_i32"] = np.asarray(
      [int(mk["cooldownTime"]), int(mk["beamLength"]), int(mk["beamRadius"]),
       hit_names.index("mine"), int(ee["minimumFramesPerEpisode"]), int(ee["intervalLength"])] +
      [v for o in by_type for v in (int(o["minNumMiners"]), int(o["miningWindow"]))], np.int32)
  # per player: reward for mining / extracting ore type k under the avatar's role
  rew = []
  for av in t["_avatars"][:P]:
    kw = _get_component(av, "MineBeam")["kwargs"]
    assert (int(kw["cooldownTime"]), int(kw["beamLength"]), int(kw["beamRadius"])) == (
        int(mk["cooldownTime"]), int(mk["beamLength"]), int(mk["beamRadius"]))
    role = kw["agentRole"]
    rew += [float(kw["roleRewardForMining"][role][k]) for k in range(K)]
    rew += [float(kw["roleRewardForExtracting"][role][k]) for k in range(K)]
  t["cm_f64"] = np.asarray(rew + [float(r) for r in rk["liveRates"]] +
                           [float(ee["probabilityTerminationPerInterval"])], np.float64)
  t["cm_thr"] = np.asarray([prob_threshold(float(r)) for r in rk["liveRates"]] +
                           [prob_threshold(float(ee["probabilityTerminationPerInterval"]))],
                           np.uint64)
  return {k: v for k, v in t.items() if not k.startswith("_")}


def lower_gift_refinements(settings: Mapping[str, Any], action_set) -> Dict[str, np.ndarray]:
  """gift_refinements: reference `configs/substrates/gift_refinements.py`,
  `lua/levels/gift_refinements/components.lua` (FixedRateRegrow :29-55 — here a
  component `update()`, not an engine-side updater —, Pickable :57-90, GiftBeam :92-237,
  Inventory :239-353, TokenTracker :355-393)."""
  t = lower_common(settings)
  hdr = t["hdr"]
  hdr[HDR_SUBSTRATE] = SUBSTRATE_IDS["gift_refinements"]
  W, P = int(hdr[HDR_W]), int(hdr[HDR_P])
  sid = t["_state_ids"]
  objs = t["objects"]
  assert t["_action_names"] == ("move", "turn", "refineAndGift", "consumeTokens")
  t["action_table"] = _action_table(action_set, t["_action_names"])
  hdr[HDR_NACT] = len(action_set)
  token = settings["simulation"]["prefabs"]["token"]
  pk = _get_component(token, "Pickable")["kwargs"]
  rk = _get_component(token, "FixedRateRegrow")["kwargs"]
  assert (pk["liveState"], pk["waitState"]) == (rk["liveState"], rk["waitState"])
  ee = _get_component(settings["simulation"]["scene"],
                      "StochasticIntervalEpisodeEnding")["kwargs"]
  t["token_cells"] = _cells_of_kind(objs, KIND_TOKEN, W)
  t["gr_states"] = np.asarray([sid[(id(token), rk["waitState"])],
                               sid[(id(token), rk["liveState"])]], np.int32)
  av0 = t["_avatars"][0]
  gk = _get_component(av0, "GiftBeam")["kwargs"]
  ik = _get_component(av0, "Inventory")["kwargs"]
  K = int(ik["numTokenTypes"])
  assert 1 <= K <= 3, "an avatar's inventory is kept in three bytes"
  assert 1 <= int(ik["capacityPerType"]) <= 15, "a 'gift' event row carries the new count in four bits"
  assert P <= 16
  hit_names = [h[0] for h in t["_hits"]]
  t["gr_i32"] = np.asarray(
      [int(gk["cooldownTime"]), int(gk["beamLength"]), int(gk["beamRadius"]),
       hit_names.index("gift"), int(ee["minimumFramesPerEpisode"]), int(ee["intervalLength"]),
       int(ik["capacityPerType"]), K, int(gk["giftMultiplier"]),
       int(ik.get("consumptionCooldown", 0))], np.int32)
  assert int(gk["cooldownTime"]) >= 1 and int(gk["giftMultiplier"]) >= 1
  # per player: roleRewardForGifting[agentRole] (paid for every hit) and that times
  # successfulGiftReward (paid when the gift was refined)
  rew, roles = [], []
  for av in t["_avatars"][:P]:
    kw = _get_component(av, "GiftBeam")["kwargs"]
    assert {k: v for k, v in kw.items() if k != "agentRole"} == {
        k: v for k, v in gk.items() if k != "agentRole"}
    assert _get_component(av, "Inventory")["kwargs"] == ik
    metrics = _get_component(av, "AvatarMetricReporter")["kwargs"]["metrics"]
    assert [(m["name"], m["component"], m["variable"]) for m in metrics] == [
        ("INVENTORY", "Inventory", "inventory")]
    role = kw["agentRole"]
    # (a role the table does not name pays nothing per hit — and the Lua fails on its
    # first refined gift, nil * successfulGiftReward, components.lua:155: refused here)
    assert role in kw["roleRewardForGifting"], f"agentRole {role!r} has no gifting reward"
    amount = float(kw["roleRewardForGifting"][role])
    rew += [amount, amount * float(kw["successfulGiftReward"])]
    roles.append(str(role))
  # the `gift` event names both avatars' roles (components.lua:174-181): host-side strings
  t["agent_roles"] = np.frombuffer(b"".join(r.encode() + b"\0" for r in roles), np.uint8)
  t["gr_f64"] = np.asarray(rew + [float(pk["rewardForPicking"]), float(rk["regrowRate"]),
                                  float(ee["probabilityTerminationPerInterval"])], np.float64)
  t["gr_thr"] = np.asarray([prob_threshold(float(rk["regrowRate"])),
                            prob_threshold(float(ee["probabilityTerminationPerInterval"]))],
                           np.uint64)
  return {k: v for k, v in t.items() if not k.startswith("_")}


MUSHROOM_TYPES = ("fullInternalityZeroExternality", "halfInternalityHalfExternality",
                  "zeroInternalityFullExternality", "negativeInternalityNegativeExternality")


def lower_externality_mushrooms(settings: Mapping[str, Any], action_set) -> Dict[str, np.ndarray]:
  """externality_mushrooms__dense: reference `configs/substrates/externality_mushrooms.py` (+ the
  map of `externality_mushrooms__dense.py`), `lua/levels/externality_mushrooms/components.lua`
  (MushroomEating :30-153, MushroomGrowable :155-194, MushroomRegrowth :196-255, Destroyable
  :257-306, Perishable :308-335), `lua/modules/avatar_library.lua:948-1121`
  (GraduatedSanctionsMarking, here with avatars that come back).  The reward rule of a mushroom
  goes by its state's NAME in the Lua (:73-104): the four names are fixed, the numbers are
  the pack's."""
  t = lower_common(settings)
  hdr = t["hdr"]
  hdr[HDR_SUBSTRATE] = SUBSTRATE_IDS["externality_mushrooms"]
  W, P = int(hdr[HDR_W]), int(hdr[HDR_P])
  sid = t["_state_ids"]
  objs = t["objects"]
  assert t["_action_names"] == ("move", "turn", "fireZap")
  t["action_table"] = _action_table(action_set, t["_action_names"])
  hdr[HDR_NACT] = len(action_set)
  av0 = t["_avatars"][0]
  t["zapper_i32"], t["zapper_f64"] = _zapper_tables(av0)
  assert not t["zapper_i32"][4], "GraduatedSanctionsMarking removes, not Zapper"
  assert 1 <= int(t["zapper_i32"][3]) <= 1 << 20

  sim = settings["simulation"]
  mushrooms = [o for o, _, _ in t["_objects"] if _get_component(o, "MushroomEating")]
  m0 = mushrooms[0]
  ek = _get_component(m0, "MushroomEating")["kwargs"]
  dk = _get_component(m0, "Destroyable")["kwargs"]
  pk = _get_component(m0, "Perishable")["kwargs"]
  for m in mushrooms:   # the prefabs differ in their initial state only
    for name in ("MushroomEating", "Destroyable", "Perishable"):
      assert _get_component(m, name)["kwargs"] == _get_component(m0, name)["kwargs"]
    assert all(sid[(id(m), s)] == sid[(id(m0), s)] for s in MUSHROOM_TYPES + ("wait",))
  assert tuple(ek["liveStates"]) == MUSHROOM_TYPES and ek.get("waitState", "wait") == "wait"
  assert dk["waitState"] == "wait" and pk["waitState"] == "wait"
  assert int(dk["initialHealth"]) == 1, "a zap destroys a mushroom (the kernel keeps no health)"
  rk = _get_component(sim["scene"], "MushroomRegrowth")["kwargs"]
  ee = _get_component(sim["scene"], "StochasticIntervalEpisodeEnding")["kwargs"]
  marking = [o for o in sim["gameObjects"] if _get_component(o, "GraduatedSanctionsMarking")]
  assert len(marking) == P
  gk = _get_component(marking[0], "GraduatedSanctionsMarking")["kwargs"]
  assert gk["hitName"] == "zapHit" and int(gk.get("initialLevel", 1)) == 1
  logic = gk["hitLogic"]
  assert len(logic) == 2
  for i, m in enumerate(marking):
    kw = _get_component(m, "GraduatedSanctionsMarking")["kwargs"]
    assert int(kw["playerIndex"]) == i + 1
    assert {k: v for k, v in kw.items() if k != "playerIndex"} == {
        k: v for k, v in gk.items() if k != "playerIndex"}
    assert sid[(id(m), "level_1")] == sid[(id(marking[0]), "level_1")]

  t["mushroom_cells"] = _cells_of_kind(objs, KIND_MUSHROOM, W)
  n_site = len(t["mushroom_cells"])
  assert n_site == len(mushrooms) and len(set(t["mushroom_cells"].tolist())) == n_site
  assert n_site <= 256, "four sites per lane; getGroupShuffledWithProbability draws by eater * 256 + site"
  live0 = sum(_get_component(m, "StateManager")["kwargs"]["initialState"] != "wait" for m in mushrooms)
  t["em_states"] = np.asarray(
      [sid[(id(m0), s)] for s in MUSHROOM_TYPES] + [sid[(id(m0), "wait")],
       sid[(id(marking[0]), "level_1")], sid[(id(marking[0]), "level_2")],
       sid[(id(marking[0]), gk["waitState"])]], np.int32)
  max_frames = int(settings.get("maxEpisodeLengthFrames", 3600))
  perish = []
  for s in MUSHROOM_TYPES:
    d = int(pk["delayPerState"][s])
    # (an age is kept in a byte; 1e7 frames never pass in an episode)
    assert 1 <= d <= 254 or d > max_frames + 1, "Perishable delay"
    perish.append(d if d <= 254 else 1 << 30)
  destroy = []
  for s in MUSHROOM_TYPES:
    rule = (ek.get("destroyOnEating") or {}).get(s)
    destroy.append(MUSHROOM_TYPES.index(rule["typeToDestroy"]) if rule else -1)
  hit_names = [h[0] for h in t["_hits"]]
  t["em_i32"] = np.asarray(
      [int(rk.get("minPotentialMushrooms", 10)), int(dk["initialHealth"]), int(gk["recoveryTime"]),
       len(logic), int(ee["minimumFramesPerEpisode"]), int(ee["intervalLength"]),
       hit_names.index("zapHit"), int(live0)] +
      [int(ek["numSporesReleasedWhenEaten"][s]) for s in MUSHROOM_TYPES] +
      [int(ek["digestionTimes"][s]) for s in MUSHROOM_TYPES] + perish + destroy +
      [v for lv in logic for v in (int(lv["levelIncrement"]), int(lv.get("freeze") or 0),
                                    int(bool(lv.get("remove", False))))], np.int32)
  assert all(0 <= int(ek["numSporesReleasedWhenEaten"][s]) <= 4 for s in MUSHROOM_TYPES)
  assert all(0 <= int(ek["digestionTimes"][s]) <= 255 for s in MUSHROOM_TYPES)
  t["em_f64"] = np.asarray(
      [float(ek["totalReward"][s]) for s in MUSHROOM_TYPES] +
      [v for lv in logic for v in (float(lv["sourceReward"]), float(lv["targetReward"]))],
      np.float64)
  probs = rk["mushroomsToProbabilities"]
  t["em_thr"] = np.asarray(
      [prob_threshold(float(probs[e][m])) for e in MUSHROOM_TYPES for m in MUSHROOM_TYPES] +
      [prob_threshold(float(((ek.get("destroyOnEating") or {}).get(s) or {}).get("percentToDestroy", 0.0)))
       for s in MUSHROOM_TYPES] +
      [prob_threshold(float(ee["probabilityTerminationPerInterval"]))], np.uint64)
  return {k: v for k, v in t.items() if not k.startswith("_")}


_EMPTY_SCENE = {"name": "scene", "components": [
    {"component": "StateManager",
     "kwargs": {"initialState": "scene", "stateConfigs": [{"state": "scene"}]}},
    {"component": "Transform"}]}


def lower_collaborative_cooking(settings: Mapping[str, Any], action_set) -> Dict[str, np.ndarray]:
  """collaborative_cooking__*: reference `configs/substrates/collaborative_cooking.py` (+ the
  layout modules), `lua/levels/collaborative_cooking/components.lua` (InteractBeam :29-113,
  Container :116-181, Inventory :184-277, Receiver :280-333, CookingPot :336-474,
  LoadingBarVisualiser :477-517).  The settings name no scene object: an empty one stands in
  as object 0.  An avatar's inventory piece turns with its avatar (avatar_library.lua:
  162-164,179-181) and its sprites rotate: the pack holds one pseudo-state per (item, facing)
  for the `<item>_offset` states (`state_orient`), as for oriented beam sprites."""
  sim = dict(settings["simulation"])
  sim.setdefault("scene", _EMPTY_SCENE)
  settings = dict(settings, simulation=sim)
  t = lower_common(settings)
  hdr = t["hdr"]
  hdr[HDR_SUBSTRATE] = SUBSTRATE_IDS["collaborative_cooking"]
  W, P = int(hdr[HDR_W]), int(hdr[HDR_P])
  sid = t["_state_ids"]
  objs = t["objects"]
  assert t["_action_names"] == ("move", "turn", "interact")
  t["action_table"] = _action_table(action_set, t["_action_names"])
  hdr[HDR_NACT] = len(action_set)
  prefabs = sim["prefabs"]
  items = ["empty", "tomato", "dish", "soup"]
  inv = prefabs["inventory"]
  ik = _get_component(inv, "Inventory")["kwargs"]
  assert (ik["emptyState"], ik["waitState"]) == ("empty", "wait")
  names = [c["state"] for c in _get_component(inv, "StateManager")["kwargs"]["stateConfigs"]]
  assert names == ["wait"] + items + [i + "_offset" for i in items], "Inventory:getHeldItem strips '_offset'"
  # every inventory object is a copy of the prefab (one set of state ids)
  inv_objs = [o for o in sim["gameObjects"] if _get_component(o, "Inventory") and not _get_component(o, "Avatar")]
  some = inv_objs[0]
  plain = [sid[(id(some), i)] for i in items]
  offs = [sid[(id(some), i + "_offset")] for i in items]
  for o in inv_objs:
    assert [sid[(id(o), n)] for n in names] == [sid[(id(some), n)] for n in names]
  # facing pseudo-states of the offset states (facing N is the state itself)
  layer_ids, sprite_ids = t["state_layer"].tolist(), t["state_sprite"].tolist()
  groups, contact = t["state_groups"].tolist(), t["state_contact"].tolist()
  orient, block = t["state_orient"].tolist(), t["state_hit_block"].tolist()
  snames = bytes(t["state_names"]).split(b"\0")[:-1]
  dirs = [list(offs)]
  for d in (1, 2, 3):
    row = []
    for k, s0 in enumerate(offs):
      row.append(len(layer_ids))
      layer_ids.append(layer_ids[s0]); sprite_ids.append(sprite_ids[s0]); groups.append(groups[s0])
      contact.append(contact[s0]); orient.append(d); block.append(block[s0])
      snames.append(snames[s0] + b"." + COMPASS[d].encode())
    dirs.append(row)
  assert len(layer_ids) <= 255
  t["state_layer"] = np.asarray(layer_ids, t["state_layer"].dtype)
  t["state_sprite"] = np.asarray(sprite_ids, t["state_sprite"].dtype)
  t["state_groups"] = np.asarray(groups, t["state_groups"].dtype)
  t["state_contact"] = np.asarray(contact, t["state_contact"].dtype)
  t["state_orient"] = np.asarray(orient, t["state_orient"].dtype)
  t["state_hit_block"] = np.asarray(block, t["state_hit_block"].dtype)
  t["state_names"] = np.frombuffer(b"\0".join(snames) + b"\0", np.uint8).copy()
  hdr[HDR_NSTATES] = len(layer_ids)
  assert plain == list(range(plain[0], plain[0] + 4)) and offs == list(range(offs[0], offs[0] + 4))
  assert [s for row in dirs[1:] for s in row] == list(range(dirs[1][0], dirs[1][0] + 12))
  t["cc_inv_states"] = np.asarray([sid[(id(some), "wait")], plain[0], offs[0], dirs[1][0]], np.int32)

  # containers (counters, dispensers) in creation order; the inventory over each
  ct = [(o, x, y) for o, x, y in t["_objects"] if _kind_of(o) == KIND_CONTAINER]
  t["cc_container_cells"] = np.asarray([y * W + x for _, x, y in ct], np.int32)
  ck = [_get_component(o, "Container").get("kwargs") or {} for o, _, _ in ct]
  t["cc_container_i32"] = np.asarray(
      [[items.index(k.get("startingItem", "empty")), int(bool(k.get("infinite", False)))] for k in ck],
      np.int32).reshape(-1, 2)
  over = {(x, y) for o, x, y in t["_objects"] if _kind_of(o) == KIND_INVENTORY
          and int(_get_component(o, "Inventory")["kwargs"]["playerIndex"]) == -1}
  assert over == {(x, y) for _, x, y in ct}, "one inventory over every container, none elsewhere"
  rc = [(o, x, y) for o, x, y in t["_objects"] if _kind_of(o) == KIND_RECEIVER]
  t["cc_receiver_cells"] = np.asarray([y * W + x for _, x, y in rc], np.int32)
  rk = [_get_component(o, "Receiver")["kwargs"] for o, _, _ in rc]
  t["cc_receiver_i32"] = np.asarray(
      [[items.index(k.get("acceptedItems", "onion")), int(bool(k.get("globalReward", False)))] for k in rk],
      np.int32).reshape(-1, 2)
  t["cc_receiver_f64"] = np.asarray([float(k.get("reward", 0)) for k in rk], np.float64)
  pots = [(o, x, y) for o, x, y in t["_objects"] if _kind_of(o) == KIND_POT]
  t["cc_pot_cells"] = np.asarray([y * W + x for _, x, y in pots], np.int32)
  bars = {(x, y) for o, x, y in t["_objects"] if _kind_of(o) == KIND_LOADING_BAR}
  assert bars == {(x, y) for _, x, y in pots}, "one loading bar over every pot, none elsewhere"
  pot = prefabs["cooking_pot"]
  pk = _get_component(pot, "CookingPot")["kwargs"]
  assert list(pk.get("acceptedItems", ["onion", "tomato"])) == ["tomato"], "one ingredient: a pot's content is a count"
  custom = list(pk["customStateNames"])

  def first_match(pattern):   # CookingPot:onHit / tickPotFn: the first custom state name holding the pattern
    return next(n for n in custom if pattern in n)
  contents = ["empty_empty_empty", "tomato_empty_empty", "tomato_tomato_empty", "tomato_tomato_tomato"]
  t["cc_pot_states"] = np.asarray([sid[(id(pot), first_match(c))] for c in contents] +
                                  [sid[(id(pot), first_match("cooked"))]], np.int32)
  bar = prefabs["loading_bar"]
  bk = _get_component(bar, "LoadingBarVisualiser")["kwargs"]
  bnames = list(bk["customStateNames"])
  assert len(bnames) == 11
  # (the loading bars are copies of the prefab: their states are found by name)
  t["cc_bar_states"] = np.asarray([sid[(bar["name"], n)] for n in bnames], np.int32)
  total = int(bk.get("totalTime", 10))
  assert total % 10 == 0, "the bar's interval (totalTime / 10) is kept as an integer"
  beams = [_get_component(av, "InteractBeam")["kwargs"] for av in t["_avatars"][:P]]
  assert all(int(b["cooldownTime"]) == int(beams[0]["cooldownTime"]) for b in beams)
  hit_names = [h[0] for h in t["_hits"]]
  t["cc_hits"] = np.asarray([hit_names.index(f"interact_{p + 1}") for p in range(P)], np.int32)
  t["cc_i32"] = np.asarray([int(beams[0]["cooldownTime"]), int(pk.get("cookingTime", 20)), total // 10],
                           np.int32)
  t["cc_f64"] = np.asarray([float(pk.get("reward", 0))], np.float64)
  # The engine's renderer takes at most 12 layers: with one interact layer per avatar the
  # nine- and six-player layouts have 16 and 13.  Layers no state is ever on (this level uses
  # neither alternateLogic, background, lowerPhysical nor superOverlay) are dropped THERE — the
  # render order of the others is kept, every pixel is the same; only a LAYER observation
  # (not one of this level's) would list fewer planes.
  L = int(hdr[HDR_L])
  if L > 12:
    used = sorted({int(l) for l in layer_ids if l >= 0})
    remap = {old: new for new, old in enumerate(used)}
    layer_ids = [remap[l] if l >= 0 else -1 for l in layer_ids]
    t["state_layer"] = np.asarray(layer_ids, t["state_layer"].dtype)
    t["init_grid"] = np.ascontiguousarray(t["init_grid"][used])
    lnames = bytes(t["layer_names"]).split(b"\0")[:-1]
    t["layer_names"] = np.frombuffer(b"\0".join(lnames[l] for l in used) + b"\0", np.uint8).copy()
    hdr[HDR_L] = len(used)
    hdr[HDR_AVATAR_LAYER] = remap[int(hdr[HDR_AVATAR_LAYER])]
    assert len(used) <= 12
  # what the object in a state is to an interact beam: 1 container, 2 dispenser (a container
  # that keeps its item), 3 receiver, 4 pot (step_cook.h COOK_KIND_*)
  kind = np.zeros(len(layer_ids), np.uint8)
  for o, _, _ in t["_objects"][1:]:
    k = _kind_of(o)
    code = {KIND_RECEIVER: 3, KIND_POT: 4}.get(k, 0)
    if k == KIND_CONTAINER:
      code = 2 if (_get_component(o, "Container").get("kwargs") or {}).get("infinite", False) else 1
    if code:
      for cfg in _get_component(o, "StateManager")["kwargs"]["stateConfigs"]:
        kind[sid[(id(o), cfg["state"])]] = code
  t["cc_state_kind"] = kind
  assert len({(int(a), int(b), float(r)) for (a, b), r in zip(t["cc_receiver_i32"], t["cc_receiver_f64"])}) <= 1, (
      "one kind of receiver (its constants are the level's)")
  return {k: v for k, v in t.items() if not k.startswith("_")}


def lower_the_matrix(settings: Mapping[str, Any], action_set) -> Dict[str, np.ndarray]:
  """*_in_the_matrix: reference `configs/substrates/<game>_in_the_matrix__*.py` +
  `the_matrix.py`, `lua/levels/the_matrix/components.lua` (TheMatrix, Resource,
  Destroyable, GameInteractionZapper, Taste, InteractionTaste, DyadicRole,
  SpawnResourcesWhenAllPlayersZapped, ReadyToInteractMarker),
  `lua/modules/avatar_library.lua:884-945` (AvatarConnector)."""
  t = lower_common(settings)
  hdr = t["hdr"]
  hdr[HDR_SUBSTRATE] = SUBSTRATE_IDS["the_matrix"]
  W, P = int(hdr[HDR_W]), int(hdr[HDR_P])
  sid = t["_state_ids"]
  objs = t["objects"]
  assert t["_action_names"] == ("move", "turn", "interact")
  t["action_table"] = _action_table(action_set, t["_action_names"])
  hdr[HDR_NACT] = len(action_set)
  sim = settings["simulation"]
  scene = sim["scene"]
  mk = _get_component(scene, "TheMatrix")["kwargs"]
  ee = _get_component(scene, "StochasticIntervalEpisodeEnding")
  ee = ee["kwargs"] if ee else None
  matrix = np.asarray(mk["matrix"], np.float64)
  R = matrix.shape[0]
  assert matrix.shape == (R, R) and 1 <= R <= 3
  # TheMatrix.__init__ (components.lua:209-216): the column player's matrix
  # defaults to the transpose of the row player's
  col = (np.asarray(mk["columnPlayerMatrix"], np.float64)
         if mk.get("columnPlayerMatrix") is not None else matrix.T.copy())
  assert col.shape == (R, R)
  intervals = [(float(a), float(b)) for a, b in mk["resultIndicatorColorIntervals"]]
  assert 1 <= len(intervals) <= 5

  avatars = t["_avatars"]
  gk = _get_component(avatars[0], "GameInteractionZapper")["kwargs"]
  for av in avatars:   # one set of zapper rules (per-player constants: mx_player_*)
    assert _get_component(av, "GameInteractionZapper")["kwargs"] == gk
    assert not _get_component(av, "Avatar")["kwargs"].get("skipWaitStateRewards", True)
    assert float(_get_component(av, "Avatar")["kwargs"].get("speed", 1.0)) == 1.0
  assert int(gk["numResources"]) == R
  spawn_all = [bool(_get_component(av, "SpawnResourcesWhenAllPlayersZapped"))
               for av in avatars]
  assert len(set(spawn_all)) == 1

  # resources: one prefab per class; the sites in object (= creation) order
  res = {}
  for name, pf in sim["prefabs"].items():
    rc = _get_component(pf, "Resource")
    if rc:
      res[int(rc["kwargs"]["resourceClass"])] = pf
  assert sorted(res) == list(range(1, R + 1))
  rk = _get_component(res[1], "Resource")["kwargs"]
  dk = _get_component(res[1], "Destroyable")["kwargs"]
  for k, pf in res.items():
    r2 = _get_component(pf, "Resource")["kwargs"]
    d2 = _get_component(pf, "Destroyable")["kwargs"]
    assert (r2["regenerationRate"], r2["regenerationDelay"]) == (
        rk["regenerationRate"], rk["regenerationDelay"])
    assert d2["initialHealth"] == dk["initialHealth"] and d2["waitState"] == r2["waitState"]
  regen_rate, regen_delay = float(rk["regenerationRate"]), int(rk["regenerationDelay"])
  assert 1 <= int(dk["initialHealth"]) <= 3
  assert regen_delay <= 250 or regen_rate == 0.0   # site ages are bytes
  site_rows = [i for i in range(len(objs)) if objs[i, 0] == KIND_RESOURCE]
  t["resource_cells"] = np.asarray([objs[i, 2] * W + objs[i, 1] for i in site_rows], np.int32)
  cls = []
  for i in site_rows:
    rc = _get_component(t["_objects"][i][0], "Resource")["kwargs"]
    cls.append(int(rc["resourceClass"]))
  t["resource_class"] = np.asarray(cls, np.int32)

  markers = [o for o in sim["gameObjects"] if _get_component(o, "ReadyToInteractMarker")]
  assert len(markers) == len(avatars)
  for i, m in enumerate(markers):
    ck = _get_component(m, "AvatarConnector")["kwargs"]
    assert int(ck["playerIndex"]) == i + 1
    assert int(_get_component(m, "ReadyToInteractMarker")["kwargs"]["playerIndex"]) == i + 1
    assert (ck["aliveState"], ck["waitState"]) == ("notReady", "avatarMarkingWait")
    assert sid[(id(m), "ready")] == sid[(id(markers[0]), "ready")]   # shared states
  m0 = markers[0]
  t["mx_states"] = np.asarray(
      [sid[(id(m0), "avatarMarkingWait")], sid[(id(m0), "ready")], sid[(id(m0), "notReady")]] +
      [sid[(id(m0), f"resultIndicatorColor{k + 1}")] for k in range(5)] +
      [v for k in range(1, R + 1) for v in (
          sid[(id(res[k]), _get_component(res[k], "Resource")["kwargs"]["visibleType"])],
          sid[(id(res[k]), _get_component(res[k], "Resource")["kwargs"]["waitState"])])],
      np.int32)
  hit_names = [h[0] for h in t["_hits"]]
  t["mx_i32"] = np.asarray([
      R, int(gk["cooldownTime"]), int(gk["beamLength"]), int(gk["beamRadius"]),
      int(gk["framesTillRespawn"]), int(gk.get("freezeOnInteraction", 0)),
      int(bool(gk.get("endEpisodeOnFirstInteraction", False))),
      int(bool(gk.get("reset_winner_inventory", False))),
      int(bool(gk.get("reset_loser_inventory", True))),
      int(bool(gk.get("losingPlayerDies", True))),
      int(bool(gk.get("winningPlayerDies", False))),
      int(bool(mk.get("zeroInitialInventory", False))),
      int(bool(mk.get("randomTieBreaking", False))),
      int(bool(mk.get("disallowUnreadyInteractions", False))),
      int(ee is not None),
      int(ee["minimumFramesPerEpisode"]) if ee else 0,
      int(ee["intervalLength"]) if ee else 1,
      regen_delay, int(dk["initialHealth"]), len(intervals), int(spawn_all[0]),
      hit_names.index("gameInteraction"),
  ], np.int32)
  ee_p = float(ee["probabilityTerminationPerInterval"]) if ee else 0.0
  t["mx_f64"] = np.asarray(
      [float(gk.get("rewardFloor", -1e6)), float(gk.get("rewardMultiplier", 1.0)),
       float(gk.get("rewardFromZappingUnreadyPlayer", 0)), regen_rate, ee_p] +
      [float(v) for v in matrix.reshape(-1)] + [float(v) for v in col.reshape(-1)] +
      [v for iv in intervals for v in iv], np.float64)
  t["mx_thr"] = np.asarray([prob_threshold(regen_rate), prob_threshold(ee_p)], np.uint64)
  pi, pf = [], []
  for av in avatars:
    ta = _get_component(av, "Taste")
    ta = ta["kwargs"] if ta else {"mostTastyResourceClass": -1}
    it = _get_component(av, "InteractionTaste")
    ro = _get_component(av, "DyadicRole")
    # no InteractionTaste component: the rewards are delivered as they are
    # (components.lua:531-538) = a component whose class is -1
    itk = it["kwargs"] if it else {}
    pi += [int(ta["mostTastyResourceClass"]), int(itk.get("mostTastyResourceClass", -1)),
           int(bool(itk.get("zeroDefaultInteractionReward", False))),
           int(bool(ro["kwargs"]["rowPlayer"])) if ro else -1]
    pf += [float(ta.get("mostTastyReward", 1)), float(ta.get("defaultTastinessReward", 0)),
           float(itk.get("extraReward", 0)), 0.0]
  t["mx_player_i32"] = np.asarray(pi, np.int32).reshape(-1, 4)
  t["mx_player_f64"] = np.asarray(pf, np.float64).reshape(-1, 4)
  return {k: v for k, v in t.items() if not k.startswith("_")}


def coins_with_every_map(settings, mod, config):
  """coins.py draws the map size inside build() (get_ascii_map, :45-82): every
  environment has its own map, for all its episodes.  Returns the settings with
  all the maps the generator can draw — outcome (w - min_width) * n_heights +
  (h - min_height), produced by the config's own generator with its two randint
  draws fixed — as the alternatives of one per-world choice."""
  maps = []
  real = mod.random.randint
  try:
    for w in range(config.min_width, config.max_width + 1):
      for h in range(config.min_height, config.max_height + 1):
        draws = iter((w, h))
        mod.random.randint = lambda a, b: next(draws)
        maps.append(mod.get_ascii_map(config.min_width, config.max_width,
                                      config.min_height, config.max_height))
  finally:
    mod.random.randint = real
  out = dict(settings)
  out["simulation"] = dict(settings["simulation"])
  out["simulation"]["mapAlternatives"] = maps
  out["simulation"]["mapChoiceScope"] = "world"
  # ... and the two coin colours (coins.py:500: random.sample(COIN_PALETTES, k=2),
  # 20 ordered pairs; player 1 and coin type A wear the first, player 2 and type B
  # the second).  The coin prefab and the avatars get a state and a sprite per
  # colour — the config's own shapes with COIN_PALETTES[colour], as get_coin /
  # build_avatar_objects make them for the drawn pair — and a world uses the pair
  # it drew (lower_coins: co_colour_*; the instance's own pair stays in the pack's
  # ordinary tables).
  import copy
  colours = list(mod.COIN_PALETTES)
  prefabs = dict(out["simulation"]["prefabs"])
  coin = copy.deepcopy(prefabs["coin"])
  sm = _get_component(coin, "StateManager")["kwargs"]
  ap = _get_component(coin, "Appearance")["kwargs"]
  for c in colours:
    if c not in [cfg["state"] for cfg in sm["stateConfigs"]]:
      sm["stateConfigs"].append({"state": c, "layer": "superOverlay", "sprite": c})
      ap["spriteNames"] = list(ap["spriteNames"]) + [c]
      ap["spriteShapes"] = list(ap["spriteShapes"]) + [ap["spriteShapes"][0]]
      ap["palettes"] = list(ap["palettes"]) + [mod.COIN_PALETTES[c]]
      ap["noRotates"] = list(ap["noRotates"]) + [ap["noRotates"][0]]
  prefabs["coin"] = coin
  out["simulation"]["prefabs"] = prefabs
  avatars = []
  for av in out["simulation"]["gameObjects"]:
    av = copy.deepcopy(av)
    sm = _get_component(av, "StateManager")["kwargs"]
    ap = _get_component(av, "Appearance")["kwargs"]
    alive = next(cfg for cfg in sm["stateConfigs"]
                 if cfg["state"] == _get_component(av, "Avatar")["kwargs"]["aliveState"])
    for c in colours:
      sprite = f"{alive['sprite']}_{c}"
      sm["stateConfigs"].append(dict(alive, state=f"{alive['state']}_{c}", sprite=sprite))
      ap["spriteNames"] = list(ap["spriteNames"]) + [sprite]
      ap["spriteShapes"] = list(ap["spriteShapes"]) + [ap["spriteShapes"][0]]
      ap["palettes"] = list(ap["palettes"]) + [mod.COIN_PALETTES[c]]
      ap["noRotates"] = list(ap["noRotates"]) + [ap["noRotates"][0]]
    avatars.append(av)
  out["simulation"]["gameObjects"] = avatars
  out["simulation"]["coinColours"] = colours
  return out


ROLE_TABLES = ("sprite_rgba", "mx_player_i32", "mx_player_f64")


def add_role_tables(base: Dict[str, np.ndarray], default_roles: Sequence[str],
                    per_role: Mapping[str, Dict[str, np.ndarray]]) -> None:
  """A substrate whose config has more than one valid role (bach_or_stravinsky:
  `create_avatar_objects(roles)`, bach_or_stravinsky_in_the_matrix__repeated.py:
  473-497) builds per-player constants from the role of each player.  `base` is
  the pack lowered for `default_roles`, `per_role[r]` the same substrate lowered
  with EVERY player in role r.  Whatever differs must be per-player data — the
  avatar's sprite (palette) and its row of mx_player_* (DyadicRole.rowPlayer, Taste)
  — and goes into the pack per (role, player), so that an engine can be created
  for any assignment (MpConfig.roles):
    role_names      NUL-separated, sorted
    role_default    i32 [P]            the assignment the rest of the pack holds
    role_sprite     i32 [P]            sprite index of player p's avatar
    role_rgba       u8  [n_roles][P][4 * S * S * 4]   that sprite under role r
    role_player_i32 i32 [n_roles][P][4],  role_player_f64 f64 [n_roles][P][4]"""
  names = sorted(per_role)
  P = int(base["hdr"][HDR_P])
  S = int(base["hdr"][HDR_SPRITE])
  block = 4 * S * S * 4
  alive = base["avatar_alive_state"]
  sprite = np.asarray([base["state_sprite"][alive[p]] for p in range(P)], np.int32)
  assert len(set(sprite.tolist())) == P and (sprite >= 0).all()
  rgba = np.zeros((len(names), P, block), np.uint8)
  pi = np.zeros((len(names), P, 4), np.int32)
  pf = np.zeros((len(names), P, 4), np.float64)
  for r, role in enumerate(names):
    t = per_role[role]
    assert set(t) == set(base)
    for k in base:
      if k in ROLE_TABLES:
        continue
      assert t[k].shape == base[k].shape and np.array_equal(t[k], base[k]), (
          f"role {role!r} changes table {k!r}: not per-player data")
    # outside the avatars' own sprites the atlas must not depend on the roles
    mask = np.ones(base["sprite_rgba"].size, bool)
    for p in range(P):
      mask[sprite[p] * block:(sprite[p] + 1) * block] = False
    assert np.array_equal(t["sprite_rgba"].reshape(-1)[mask],
                          base["sprite_rgba"].reshape(-1)[mask])
    for p in range(P):
      rgba[r, p] = t["sprite_rgba"].reshape(-1)[sprite[p] * block:(sprite[p] + 1) * block]
    pi[r] = t["mx_player_i32"].reshape(P, 4)
    pf[r] = t["mx_player_f64"].reshape(P, 4)
  base["role_names"] = np.frombuffer(b"".join(n.encode() + b"\0" for n in names), np.uint8)
  base["role_default"] = np.asarray([names.index(r) for r in default_roles], np.int32)
  base["role_sprite"] = sprite
  base["role_rgba"] = rgba
  base["role_player_i32"] = pi
  base["role_player_f64"] = pf
  # the default assignment is one of the combinations
  for p in range(P):
    r = int(base["role_default"][p])
    assert np.array_equal(rgba[r, p], base["sprite_rgba"].reshape(-1)[
        sprite[p] * block:(sprite[p] + 1) * block])
    assert np.array_equal(pi[r, p], base["mx_player_i32"].reshape(P, 4)[p])
    assert np.array_equal(pf[r, p], base["mx_player_f64"].reshape(P, 4)[p])


def apply_roles(tables: Dict[str, np.ndarray], roles: Sequence[int]) -> Dict[str, np.ndarray]:
  """The pack an assignment of role indices stands for (what mp_create does to
  its copy of the pack; tests use it to hand the oracle the same world)."""
  out = {k: v.copy() for k, v in tables.items()}
  P = int(out["hdr"][HDR_P])
  S = int(out["hdr"][HDR_SPRITE])
  block = 4 * S * S * 4
  flat = out["sprite_rgba"].reshape(-1)
  pi = out["mx_player_i32"].reshape(P, 4)
  pf = out["mx_player_f64"].reshape(P, 4)
  for p, r in enumerate(roles):
    sp = int(out["role_sprite"][p])
    flat[sp * block:(sp + 1) * block] = out["role_rgba"].reshape(-1, P, block)[r, p]
    pi[p] = out["role_player_i32"].reshape(-1, P, 4)[r, p]
    pf[p] = out["role_player_f64"].reshape(-1, P, 4)[r, p]
  return out


def lower(name: str, settings: Mapping[str, Any], action_set,
          default_players: int = 0) -> Dict[str, np.ndarray]:
  """`default_players`: what an engine runs when its caller names no player
  count (0 = all the players of `settings`)."""
  tables = _lower(name, settings, action_set)
  assert 0 <= default_players <= int(tables["hdr"][HDR_P])
  tables["hdr"][HDR_DEFAULT_P] = default_players
  return tables


def _lower(name: str, settings: Mapping[str, Any], action_set) -> Dict[str, np.ndarray]:
  level = settings["levelName"]
  check_components(settings)
  if level == "coins":
    return lower_coins(settings, action_set)
  if level == "territory":
    return lower_territory(settings, action_set)
  if level == "clean_up":
    return lower_clean_up(settings, action_set)
  if level == "commons_harvest":
    return lower_commons_harvest(settings, action_set)
  if level == "the_matrix":
    return lower_the_matrix(settings, action_set)
  if level == "coop_mining":
    return lower_coop_mining(settings, action_set)
  if level == "gift_refinements":
    return lower_gift_refinements(settings, action_set)
  if level == "collaborative_cooking":
    return lower_collaborative_cooking(settings, action_set)
  if level == "externality_mushrooms":
    return lower_externality_mushrooms(settings, action_set)
  raise NotImplementedError(f"no lowering for level {level!r} ({name})")
