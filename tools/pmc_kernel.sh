#!/bin/bash
# dev helper: SQ wave-state counters for kernels matching $1 on bench args $2
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_k; rm -rf $O; mkdir -p $O; cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU"
n=1
for set in "$SQ1" "$SQ2"; do
  timeout -k 5 120 rocprofv3 --pmc $set -d $O/$n -o r -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline $2 > $O/$n.log 2>&1
  echo "set $n rc=$?"
  n=$((n+1))
done
python3 - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$O/*/*.db")):
    db = sqlite3.connect(f)
    rows = db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    for k, c, v in rows:
        if "$1" in k: print(c, "%.3e" % v)
PY
