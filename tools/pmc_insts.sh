#!/bin/bash
# dev helper (GPU box): instructions a fused headline launch issues, by kind, and its wave cycles, for engine builds
# (tags of meltingpot_amd/lib/libmp_engine_<tag>.so; "-" = the product):  tools/pmc_insts.sh <out dir> <tag> ...
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/$1; shift; mkdir -p $O; cd /tmp
# (PMC_SET: another counter set; BENCH_ARGS: another bench config, e.g. "--obs agents")
SQ1=${PMC_SET:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"}
for tag in "$@"; do
  if [ "$tag" = "-" ]; then unset MP_ENGINE_LIB MP_BENCH_ALLOW_DEV_ENV; else export MP_ENGINE_LIB=$R/meltingpot_amd/lib/libmp_engine_$tag.so MP_BENCH_ALLOW_DEV_ENV=1; fi
  t=${tag/-/product}
  timeout -k 5 150 rocprofv3 --pmc $SQ1 -d $O/pmc_$t -o r -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-substrate-api --no-rollout-api --no-steady-state --no-configs --no-box-fill --place 1 $BENCH_ARGS > $O/pmc_$t.log 2>&1
  echo "$t rc=$?"
done
unset MP_ENGINE_LIB MP_BENCH_ALLOW_DEV_ENV
python3 - "$O" <<'PY'
import sqlite3, glob, sys
O = sys.argv[1]
for d in sorted(glob.glob(O + "/pmc_*/")):
    for f in glob.glob(d + "**/*.db", recursive=True):
        db = sqlite3.connect(f)
        rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                          "where kernel_name like '%k_frame%' group by kernel_name, counter_name").fetchall()
        print(d.rstrip('/').split('/')[-1])
        for k, c, v, n in rows:
            print(f"   {c:22s} {v:14.4e}  (avg of {n} dispatches)  {k[-70:]}")
PY
