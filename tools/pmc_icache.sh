#!/bin/bash
# dev helper (GPU box): does the frame kernel wait for its INSTRUCTIONS?  k_frame<CleanUpTables, ., 1>
# is 12.3 K instructions (~85 KB of code) against a 64 KB instruction cache shared by two CUs.
# Counter sets in separate passes (MI355X_MICROARCH.md); prints the per-dispatch averages of the
# frame kernels.  usage: tools/pmc_icache.sh [bench.py args]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_icache; rm -rf $O; mkdir -p $O; cd /tmp
S1="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
S2="SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
S3="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
n=1
for set in "$S1" "$S2" "$S3"; do
  timeout -k 5 150 rocprofv3 --pmc $set -d $O/set$n -o r -- python $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-traffic --no-substrate-api --no-rollout-api --no-steady-state --place 1 "$@" > $O/set$n.log 2>&1
  echo "set $n rc=$?"; n=$((n+1))
done
python3 - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$O/set*/*.db") + glob.glob("$O/set*/*/*.db")):
    db = sqlite3.connect(f)
    try:
        rows = db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                          "group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(f, e); continue
    for k, c, n, v in rows:
        if "k_frame" in k: print("%-28s %-40s n=%-4d %.4e" % (c, k[:40], n, v))
PY
