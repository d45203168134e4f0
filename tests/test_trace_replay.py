"""The trace replayer (tests/tools/replay_trace.py): a trajectory recorded with
non-default engine assumptions must be attributed to exactly those assumptions.

The input a real fit needs — a DMLab2D recording made by
tools/dump_dmlab2d_trace.py — cannot be produced here (no dmlab2d wheel), so the
recording is synthetic: the oracle itself, with switches flipped, writes a file
in the recorder's format; the replayer, which only sees the file, has to recover
the switches."""
import os
import sys

import numpy as np
import pytest

import util

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
import replay_trace  # noqa: E402


def _actions(seed, steps, players, weights):
  rng = np.random.default_rng(seed)
  return util.random_actions(rng, steps, 1, players, len(weights), weights)[:, 0]


def test_trace_file_round_trip(tmp_path, clean_up_pack):
  acts = _actions(0, 12, 7, [1] * 9)
  trace = replay_trace.record_with_oracle(clean_up_pack, 99, acts)
  path = tmp_path / "t.npz"
  np.savez_compressed(path, **trace)
  back = replay_trace.load_trace(str(path))
  assert set(back) == set(trace)
  assert back["world_rgb"].shape == (13, 168, 240, 3) and back["rgb"].shape == (13, 7, 88, 88, 3)
  assert back["actions"].shape == (12, 7) and int(back["seed"]) == 99
  # defaults reproduce a default recording, frame for frame, unmasked
  r = replay_trace.replay(clean_up_pack, back, {}, mask_random_cells=False)
  assert r.first is None and r.bad_frames == 0 and r.frames == 13


@pytest.mark.parametrize("truth", [
    {"A4_beam_marks_blocked": 0, "A2_flush_count": 128},
    {"A4_beam_marks_blocked": 1, "A2_flush_count": 1},
    {"A4_beam_marks_blocked": 0, "A2_flush_count": 1},
])
def test_fit_recovers_flipped_switches(clean_up_pack, truth):
  """Beam sprite on the blocked cell or not (A4), callbacks' events in the same
  update or the next (A2) — both visible in a fertile clean_up with beam- and
  walk-heavy play."""
  pack = util.fertile_clean_up(clean_up_pack)
  acts = _actions(5, 150, 7, [1, 6, 2, 2, 2, 2, 2, 4, 4])
  trace = replay_trace.record_with_oracle(pack, 1234, acts, truth)
  reports = replay_trace.fit(pack, trace, names=list(truth), mask_random_cells=False)
  assert len(reports) == 4
  best = reports[0]
  assert best.switches == truth and best.first is None
  # every other combination is caught, and says where
  for r in reports[1:]:
    assert r.bad_frames > 0 and r.first is not None, r.line()
    assert r.first.what in ("WORLD.RGB", "reward") or r.first.what.endswith(".RGB")
  assert "reproduces all 151 frames" in best.line()


def test_fit_recovers_the_blocked_move_reentry(clean_up_pack):
  """A3b: a move that is blocked re-fires onEnter in place (component_library.lua:
  292-309), so an avatar walking into a wall while standing on a fresh apple eats
  it.  Needs apples everywhere and walking."""
  pack = util.fertile_clean_up(clean_up_pack, max_rate=1.0)
  truth = {"A3b_blocked_move_reenters": 0}
  acts = _actions(0, 300, 7, [0, 10, 2, 2, 2, 1, 1, 0, 0])
  trace = replay_trace.record_with_oracle(pack, 1234, acts, truth)
  reports = replay_trace.fit(pack, trace, names=list(truth), mask_random_cells=False)
  assert reports[0].switches == truth and reports[0].first is None
  assert reports[1].bad_frames > 0


def test_fit_recovers_the_serial_generator(clean_up_pack):
  """A10s: a trace recorded with ONE serial mt19937_64 per world (consumed in call
  order) instead of the counter-based generator is reproduced by exactly that switch —
  spawn points, visiting orders, growth and dirt all depend on it."""
  truth = {"A10s_serial_mt19937": 1, "A4_beam_marks_blocked": 0}
  acts = _actions(3, 120, 7, [1, 4, 2, 2, 2, 2, 2, 3, 3])
  trace = replay_trace.record_with_oracle(clean_up_pack, 99, acts, truth)
  reports = replay_trace.fit(clean_up_pack, trace, names=list(truth), mask_random_cells=False)
  assert reports[0].switches == truth and reports[0].first is None
  assert all(r.bad_frames > 0 for r in reports[1:])
  # the wrong generator diverges at frame 0 (the spawn shuffle), whatever A4 is
  assert all(r.first.frame == 0 for r in reports if r.switches["A10s_serial_mt19937"] == 0)


def test_switches_a_trace_does_not_exercise_tie(clean_up_pack):
  """A recording in which nobody is ever zapped says nothing about A5 / A6: the
  fit must report the tie, not pick one."""
  acts = _actions(2, 60, 7, [1, 4, 1, 1, 1, 2, 2, 0, 3])   # no zapping
  trace = replay_trace.record_with_oracle(clean_up_pack, 5, acts, {})
  reports = replay_trace.fit(clean_up_pack, trace, mask_random_cells=False,
                             names=["A5_teleport_free_only", "A6_dead_view_black"])
  assert all(r.first is None for r in reports)


def test_fit_recovers_the_visiting_order_and_the_dead_view(clean_up_pack):
  """A1 (shuffled visiting order vs creation order: who wins a contested cell)
  and A6 (what a zapped avatar sees)."""
  truth = {"A1_shuffle_order": 0, "A6_dead_view_black": 0}
  acts = _actions(8, 200, 7, [0, 8, 2, 2, 2, 1, 1, 6, 1])
  trace = replay_trace.record_with_oracle(clean_up_pack, 77, acts, truth)
  reports = replay_trace.fit(clean_up_pack, trace, names=list(truth), mask_random_cells=False)
  assert reports[0].switches == truth and reports[0].first is None
  assert all(r.bad_frames > 0 for r in reports[1:])
  by_obs = reports[-1].first_by_observation
  assert by_obs and all(d.frame >= 0 for d in by_obs.values())


def test_teacher_forced_masked_replay(clean_up_pack):
  """The mode a DMLab2D recording needs: avatars are put where the recording has
  them before every step and draw-dependent cells are masked.  On a recording
  with a DIFFERENT seed (so spawn points, growth and animation phases all
  differ, as they would against mt19937_64) the true switches still give the
  fewest diverging frames."""
  truth = {"A4_beam_marks_blocked": 0}
  acts = _actions(3, 80, 7, [1, 3, 1, 1, 1, 2, 2, 6, 6])
  trace = replay_trace.record_with_oracle(clean_up_pack, 4242, acts, truth)
  trace["seed"] = np.int64(777)   # the replayer's generator is not the recorder's
  reports = replay_trace.fit(clean_up_pack, trace, names=list(truth))
  assert reports[0].switches == truth
  assert reports[0].bad_frames < reports[1].bad_frames


# ---------------------------------------------------------------- the recorder half
# tools/dump_dmlab2d_trace.py is written for a machine that has the dmlab2d wheel.
# Here it runs UNMODIFIED against tests/tools/dmlab2d_standin.py: through the
# reference's own meltingpot.substrate.get_config, the config's build(roles,
# config), utils/substrates/builder.builder() and its reset wrapper — only
# dmlab2d.Lab2d / dmlab2d.Environment underneath are the stand-in (the product's
# lab2d_env.Environment on an oracle-backed world).

HAVE_REFERENCE = os.path.isdir("/root/reference/meltingpot")
needs_reference = pytest.mark.skipif(not HAVE_REFERENCE, reason="reference tree not present (GPU box)")


def _record(tmp_path, substrate="clean_up", players=7, steps=40, seed=99, options=None):
  import runpy
  import dmlab2d_standin
  dmlab2d_standin.install()
  dmlab2d_standin.ORACLE_OPTIONS.clear()
  dmlab2d_standin.ORACLE_OPTIONS.update(options or {})
  out = tmp_path / f"trace_{substrate}_{seed}.npz"
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  argv = sys.argv
  sys.argv = ["dump_dmlab2d_trace.py", "--substrate", substrate, "--players", str(players),
              "--steps", str(steps), "--seed", str(seed), "--out", str(out)]
  try:
    runpy.run_path(os.path.join(root, "tools", "dump_dmlab2d_trace.py"), run_name="__main__")
  finally:
    sys.argv = argv
    dmlab2d_standin.ORACLE_OPTIONS.clear()
  return replay_trace.load_trace(str(out))


@needs_reference
@pytest.mark.parametrize("substrate,players", [("clean_up", 7), ("commons_harvest__open", 5),
                                               ("territory__rooms", 9)])
def test_recorder_output_is_what_the_replayer_reads(tmp_path, substrate, players):
  """recorder -> .npz -> replayer: under the default switches the replay of the
  committed pack reproduces every frame the recorder wrote (rewards, WORLD.RGB,
  every player's RGB), unmasked."""
  from meltingpot_amd import engine
  trace = _record(tmp_path, substrate, players, steps=40, seed=99)
  assert trace["actions"].shape == (40, players) and trace["rgb"].shape[:2] == (41, players)
  r = replay_trace.replay(engine.load_pack(substrate), trace, {}, mask_random_cells=False,
                          players=players)
  assert r.first is None and r.bad_frames == 0 and r.frames == 41, r.line()


@needs_reference
def test_recorded_trace_decides_a_flipped_switch(tmp_path, clean_up_pack):
  """The recorder run against a world whose engine marks no beam sprite on the cell
  that stopped the beam (A4 = 0): the replayer's fit must name exactly that."""
  truth = {"A4_beam_marks_blocked": 0}
  trace = _record(tmp_path, "clean_up", 7, steps=250, seed=7, options=truth)
  reports = replay_trace.fit(clean_up_pack, trace, names=list(truth), mask_random_cells=False,
                             players=7)
  assert reports[0].switches == truth and reports[0].first is None
  assert all(r.first is not None for r in reports[1:])


@needs_reference
def test_recorded_trace_in_serial_generator_mode_is_recovered(tmp_path, clean_up_pack):
  """The recorder (tools/dump_dmlab2d_trace.py, through the reference's own builder on
  the stand-in dmlab2d) run against a world that draws from ONE serial mt19937_64 in
  call order — the reference's kind of generator: the replayer's fit names the switch."""
  truth = {"A10s_serial_mt19937": 1}
  trace = _record(tmp_path, "clean_up", 7, steps=120, seed=31, options=truth)
  reports = replay_trace.fit(clean_up_pack, trace, names=list(truth), mask_random_cells=False,
                             players=7)
  assert reports[0].switches == truth and reports[0].first is None
  assert reports[1].first is not None and reports[1].first.frame == 0


@needs_reference
def test_builder_seed_semantics_through_the_reference_builder():
  """utils/substrates/builder_test.py:47-75 on the stand-in world: the same
  env_seed gives the same WORLD.RGB at every reset of two separately built
  environments; consecutive episodes of one environment differ; unseeded
  environments differ from each other."""
  import dmlab2d_standin
  from meltingpot_amd import refshim
  ref = dmlab2d_standin.install()
  builder = sys.modules["meltingpot.utils.substrates.builder"]
  settings, _, _ = refshim.build_settings("commons_harvest__open", ("default",) * 4)
  for seed in (42, 12481632):
    a, b = builder.builder(settings, env_seed=seed), builder.builder(settings, env_seed=seed)
    last = None
    for episode in range(4):
      oa, ob = a.reset().observation["WORLD.RGB"], b.reset().observation["WORLD.RGB"]
      assert np.array_equal(oa, ob), episode
      assert last is None or not np.array_equal(last, oa), episode
      last = oa
    a.close(); b.close()
  a, b = builder.builder(settings), builder.builder(settings)
  assert not np.array_equal(a.reset().observation["WORLD.RGB"], b.reset().observation["WORLD.RGB"])
  a.close(); b.close()


def test_profile_summary_separates_the_timed_region(tmp_path):
  """tools/rocprof_summary.py --last N: a traced bench run's kernel table also counts
  the dry launches of Engine.place() (the same kernel, before the timed steps); the
  last N dispatches are the timed region."""
  import sqlite3
  import subprocess
  import sys
  db = tmp_path / "r_results.db"
  con = sqlite3.connect(db)
  con.execute("create table kernels (name text, start integer, duration integer)")
  rows = [("k_frame<CleanUp>", i * 200000, 90000) for i in range(40)]          # probe + warm-up
  rows += [("k_frame<CleanUp>", (40 + i) * 200000, 110000) for i in range(100)]  # timed
  rows += [("k_sum_counters", 10 ** 9, 7000)]
  con.executemany("insert into kernels values (?, ?, ?)", rows)
  con.commit(); con.close()
  log = tmp_path / "bench.log"
  log.write_text('noise\n{"steps": 100, "kernels_ms": {"frame": 0.1102}, "placement": {"picked": 3}}\n')
  out = tmp_path / "out.md"
  tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "rocprof_summary.py")
  subprocess.run([sys.executable, tool, "--trace", str(db), "--last", "100", "--bench-log", str(log),
                  "--out", str(out)], check=True, stdout=subprocess.DEVNULL)
  text = out.read_text()
  assert "| 140 |" in text                      # the table: every dispatch of the kernel
  assert "avg 110.00 µs" in text                # the timed region only
  assert "**110.20 µs** per launch" in text     # the traced process's own bench line
