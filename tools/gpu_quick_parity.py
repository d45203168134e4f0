#!/usr/bin/env python3
"""dev helper (GPU box): staged bring-up of the round-2 kernels, each stage
reporting instead of hanging (the frame kernel's waits are bounded)."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import util
from meltingpot_amd import engine as E

def stage(name):
  print(f"--- {name}", flush=True)

def cmp_state(eng, oracles, tag):
  grid, avat, glob = eng.dump()
  bad = 0
  for w, o in enumerate(oracles):
    og, oa, ogl = o.dump()
    if not (np.array_equal(grid[w], og) and np.array_equal(avat[w], oa) and np.array_equal(glob[w], ogl)):
      bad += 1
      if bad == 1:
        print(tag, "world", w, "differs: glob", glob[w], ogl, "avat eq", np.array_equal(avat[w], oa),
              "grid diffs", np.argwhere(grid[w] != og)[:5].tolist(), flush=True)
  return bad

def run(sub, n, steps, players=0):
  pack = E.load_pack(sub)
  eng = E.Engine(pack, n, num_players=players)
  oracles = util.make_oracles(pack, n, num_players=players)
  eng.reset()
  for o in oracles: o.reset()
  print(sub, "reset mismatches:", cmp_state(eng, oracles, "reset"), flush=True)
  rng = np.random.default_rng(1)
  acts = util.random_actions(rng, steps, n, eng.P, eng.num_actions)
  d = torch.from_numpy(acts).to(eng.device)
  bad_steps = 0
  for s in range(steps):
    eng.step(d[s])
    for w, o in enumerate(oracles): o.step(acts[s, w])
    if s % 5 == 4 or s == steps - 1:
      b = cmp_state(eng, oracles, f"step {s+1}")
      bad_steps += b > 0
      if b: break
  rew = eng.observe(E.OBS_REWARD).cpu().numpy()
  print(sub, "standalone step kernel:", "OK" if not bad_steps else "MISMATCH",
        "rewards ok:", all(np.array_equal(rew[w], o.rewards()) for w, o in enumerate(oracles)), flush=True)
  return eng, oracles, d, acts

def pixels(eng, oracles, kind, t):
  t = t.cpu().numpy()
  bad = 0
  for w, o in enumerate(oracles):
    if kind == E.OBS_WORLD_RGB:
      bad += not np.array_equal(t[w], o.render_world())
    else:
      bad += any(not np.array_equal(t[w, p], o.render_agent(p)) for p in range(o.P))
  return bad

try:
  stage("standalone step kernels")
  eng, oracles, d, acts = run("clean_up", 8, 30)
  stage("render-only frame kernel (mp_observe)")
  for kind, nm in ((E.OBS_WORLD_RGB, "WORLD.RGB"), (E.OBS_RGB, "RGB")):
    t0 = time.time()
    out = eng.observe(kind)
    try:
      eng.sync()
      print(nm, "render-only: worlds with wrong pixels:", pixels(eng, oracles, kind, out),
            "(%.2f s)" % (time.time() - t0), flush=True)
    except Exception as ex:
      print(nm, "render-only FAILED:", ex, flush=True)
  eng.close()
  stage("fused step + render")
  for kind, nm in ((E.OBS_WORLD_RGB, "WORLD.RGB"), (E.OBS_RGB, "RGB")):
    pack = E.load_pack("clean_up")
    eng = E.Engine(pack, 8)
    oracles = util.make_oracles(pack, 8)
    bound = eng.bind(kind)
    eng.reset()
    for o in oracles: o.reset()
    try:
      eng.sync()
      print(nm, "fused reset: state mismatches", cmp_state(eng, oracles, "fused reset"),
            "pixel mismatches", pixels(eng, oracles, kind, bound), flush=True)
      for s in range(20):
        eng.step(d[s])
        for w, o in enumerate(oracles): o.step(acts[s, w])
      eng.sync()
      print(nm, "fused 20 steps: state mismatches", cmp_state(eng, oracles, "fused step"),
            "pixel mismatches", pixels(eng, oracles, kind, bound), flush=True)
    except Exception as ex:
      print(nm, "fused FAILED:", ex, flush=True)
    eng.close()
  stage("other substrates, standalone")
  for sub in ("commons_harvest__open", "territory__rooms", "coins"):
    e2, _, _, _ = run(sub, 6, 25)
    e2.close()
  stage("step kernel timing (no views bound)")
  for sub, n, extra in (("clean_up", 4096, {}), ("commons_harvest__open", 4096, {}), ("territory__rooms", 8192, {})):
    eng = E.Engine(E.load_pack(sub), n)
    eng.reset()
    a = torch.randint(0, eng.num_actions, (16, n, eng.P), device=eng.device, dtype=torch.int32)
    for i in range(10): eng.step(a[i % 16])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200): eng.step(a[i % 16])
    e1.record(); torch.cuda.synchronize()
    print(sub, n, "worlds: %.1f us per step launch" % (e0.elapsed_time(e1) / 200 * 1e3), flush=True)
    eng.close()
except Exception:
  traceback.print_exc()
