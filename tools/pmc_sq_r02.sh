#!/bin/bash
# SQ wave-state counters of the frame kernel (fused) and of the step / draw kernels
# (--unfused) on one bench config: two PMC passes each, every run under a timeout.
# usage (on the GPU box): tools/pmc_sq_r02.sh <name> "<bench args>"  -> gpurun_out/sq_<name>.md
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sq_$1; rm -rf $O; mkdir -p $O; cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
SQ2="SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU"
for mode in fused unfused; do
  n=1
  for set in "$SQ1" "$SQ2"; do
    timeout -k 5 120 rocprofv3 --pmc $set -d $O/${mode}_$n -o r -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --$mode $2 > $O/${mode}_$n.log 2>&1
    echo "$mode set $n rc=$?"
    n=$((n+1))
  done
done
python3 - > $R/gpurun_out/sq_$1.md <<PY
import sqlite3, glob, collections
print("# SQ counters: $1 (bench.py $2), mean per dispatch\n")
for mode in ("fused", "unfused"):
    vals = collections.defaultdict(dict)
    for f in sorted(glob.glob("$O/%s_*/**/*.db" % mode, recursive=True)):
        db = sqlite3.connect(f)
        for k, c, v in db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
            if "k_frame" in k or "k_step" in k:
                vals[k.replace("(anonymous namespace)::", "")[:100]][c] = v
    print("## --%s\n" % mode)
    for k, cs in vals.items():
        print("### \`%s\`\n" % k)
        print("| counter | mean per dispatch | / SQ_WAVE_CYCLES |\n|---|---|---|")
        wc = cs.get("SQ_WAVE_CYCLES", 0) or 1
        for c, v in sorted(cs.items()):
            print("| %s | %.4g | %.3f |" % (c, v, v / wc))
        print()
PY
rm -rf $O
