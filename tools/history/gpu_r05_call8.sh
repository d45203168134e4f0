#!/bin/bash
# round 5, call 8: the whole GPU suite on the build that leaves the next step's orders in the
# record; that build against the previous one on the same buffers (tools/gpu_paired_ab.py;
# libmp_engine_old.so = the library of commit 7e7e50f); the KiB front against the product order
# on scattered buffers (tools/ubench/write_fronts.hip: VERDICT r04 item 6's kill criterion)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call8; mkdir -p $O
cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -14 $O/pytest.log
for cfg in "clean_up 4096 world" "clean_up 4096 both" "commons_harvest__open 4096 agents" "territory__rooms 4096 agents"; do
  NBUF=2 MAPPED=6 timeout 300 python tools/gpu_paired_ab.py $cfg old:static_pct=100 -:static_pct=100 -:static_pct=100,no_next_orders=1 old:static_pct=100 2>&1 | grep -v amdgpu.ids | tee -a $O/paired.txt
done
b=tools/ubench/write_fronts
hipcc --offload-arch=gfx950 -O3 -o $b $b.hip 2> $O/hipcc.err || tail $O/hipcc.err
for i in 1 2; do
  NBUF=2 VMM_LAYOUTS=1 VMM=2,2,2,2,2,2 VARIANTS=0,5 NO_DYN=1 NO_STEAL=1 timeout 120 $b 0 1 > $O/fronts_product_$i.md 2>&1; echo "rc $?"
  NBUF=2 VMM_LAYOUTS=1 VMM=2,2,2,2,2,2 VARIANTS=0,6,7 NO_DYN=1 NO_STEAL=1 timeout 120 $b 2 1 > $O/fronts_kib_$i.md 2>&1; echo "rc $?"
  NBUF=2 VMM_LAYOUTS=1 VMM=2,2,2,2,2,2 VARIANTS=0,6,7 NO_DYN=1 NO_STEAL=1 timeout 120 $b 3 1 > $O/fronts_4kib_$i.md 2>&1; echo "rc $?"
done
cat $O/fronts_product_1.md $O/fronts_kib_1.md $O/fronts_4kib_1.md | cut -c1-260
rm -f $b
