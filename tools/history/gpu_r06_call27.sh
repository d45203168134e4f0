#!/bin/bash
# round 6, call 27: the whole GPU suite and smoke on the library with the tuner's stages; the round's traced profile set
# (kernel trace + HBM traffic passes per bench config: tools/profile_round.sh r06b)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PYTHONPATH=. PYTHONUNBUFFERED=1; O=gpurun_out/r06_call27; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -12 $O/pytest_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-160
( time bash tools/profile_round.sh r06b ) > $O/profile_round.log 2>&1; echo "profile round rc=$?"; grep -E "rc=|us " $O/profile_round.log | head -20
for f in gpurun_out/prof_r06b/*.md; do echo "== $f"; head -30 $f | cut -c1-200; done
