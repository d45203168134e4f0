import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, util
import test_gpu_parity as T
from meltingpot_amd import engine as E
pack = E.load_pack("clean_up")
for tag, n, dev, placements in [
    ("dev+place 1030", 1030, {"batch_worlds": 1, "ring_batches": 8, "static_pct": 50}, 12),
    ("dev, no place", 1030, {"batch_worlds": 1, "ring_batches": 8, "static_pct": 50}, 0),
    ("dev static, place", 1030, {"batch_worlds": 1, "ring_batches": 8}, 12),
    ("stock, place", 1030, None, 12),
    ("stock, tune only", 1030, None, 1),
    ("dev static_pct only, no place", 1030, {"static_pct": 50}, 0),
    ("dev B1 NB8 pct50 world only, no place", 1030, {"batch_worlds": 1, "ring_batches": 8, "static_pct": 50}, 0),
]:
  try:
    eng = E.Engine(pack, n, dev=dev, placements=placements)
    a = eng.bind(E.OBS_RGB)
    if "world only" in tag:
      eng.unbind(E.OBS_RGB)
    b = eng.bind(E.OBS_WORLD_RGB)
    oracles = util.make_oracles(pack, n)
    eng.reset()
    for o in oracles: o.reset()
    wr = b.cpu().numpy()
    badw = [w for w, o in enumerate(oracles) if not np.array_equal(wr[w], o.render_world())]
    bada = []
    if "world only" not in tag:
      ar = a.cpu().numpy()
      bada = [w for w, o in enumerate(oracles) if any(not np.array_equal(ar[w, p], o.render_agent(p)) for p in range(7))]
    print(tag, "bad WORLD.RGB worlds:", len(badw), badw[:8], badw[-3:], "bad RGB worlds:", len(bada), bada[:8], "placement", {k: (v["candidates"], v["picked"]) for k, v in eng.placement.items()}, "faults", eng.fault_words()[:6], flush=True)
    eng.close()
  except Exception as ex:
    print(tag, "EXC", type(ex).__name__, str(ex)[:300], flush=True)
