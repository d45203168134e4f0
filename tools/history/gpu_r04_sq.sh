#!/bin/bash
# round 4 (second session): SQ instruction counters of the headline launch (is it issue-bound?)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_sq; rm -rf $O; mkdir -p $O; cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_BUSY_CYCLES"
SQ3="SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_I8 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM"
n=1
for set in "$SQ1" "$SQ2" "$SQ3"; do
  timeout -k 5 120 rocprofv3 --pmc $set -d $O/s$n -o r -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-substrate-api --place 1 ${1:-} > $O/s$n.log 2>&1
  echo "set $n rc=$?"
  n=$((n+1))
done
python3 - <<PY
import sqlite3, glob
seen = {}
for f in sorted(glob.glob("$O/*/*/*.db") + glob.glob("$O/*/*.db")):
    db = sqlite3.connect(f)
    try:
        rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(f, e); continue
    for k, c, v, n in rows:
        if "k_frame<CleanUp" in k or "k_frame<Commons" in k: seen[(k[:60], c)] = (v, n)
wc = {k: v[0] for (k, c), v in seen.items() if c == "SQ_WAVE_CYCLES"}
for (k, c), (v, n) in sorted(seen.items()):
    print(k, c, "%.3e" % v, "n=%d" % n, "/wave_cycles %.3f" % (v / wc[k]) if k in wc else "")
PY
