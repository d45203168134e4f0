#!/bin/bash
# dev helper: SQ wave-state counters of the render kernel for two engine builds (A = $1 in lib/, B = current)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sq; mkdir -p $O; cd /tmp
A=$R/meltingpot_amd/lib/$1
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU"
for tag in A B; do
  lib=""; [ $tag = A ] && lib=$A
  n=1
  for set in "$SQ1" "$SQ2"; do
    MP_ENGINE_LIB=$lib timeout -k 5 120 rocprofv3 --pmc $set -d $O/${tag}_$n -o r -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/${tag}_$n.log 2>&1
    echo "$tag set $n rc=$?"; tail -2 $O/${tag}_$n.log | cut -c1-300
    n=$((n+1))
  done
done
python3 - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$O/*/*/*.db") + glob.glob("$O/*/*.db")):
    db = sqlite3.connect(f)
    try:
        rows = db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(f, e); continue
    for k, c, v in rows:
        if "render" in k: print(f.split("/")[-3] if f.count("/")>2 else f, c, "%.3e" % v)
PY
