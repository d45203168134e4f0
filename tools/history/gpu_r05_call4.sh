#!/bin/bash
# round 5, call 4: record write-back as nt stores (A/B on the same buffers), the allocation-method
# study on another box, the address-class table again
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call4; mkdir -p $O
cd $R; export PYTHONPATH=. NBUF=4 MAPPED=2
for sub in "clean_up 4096 world" "clean_up 4096 both" "commons_harvest__open 4096 agents"; do
  n=$(echo $sub | tr ' ' '_')
  timeout 300 python tools/gpu_paired_ab.py $sub - recnt - recnt > $O/ab_$n.txt 2>&1; echo "rc $?"
  grep -v amdgpu.ids $O/ab_$n.txt
done
( cd tools/ubench && timeout 200 ./class_bw 4 x ) > $O/class_bw.md 2>&1
grep -A 22 "m = 8" $O/class_bw.md | tail -12
( time timeout 900 python tools/alloc_method_study.py --processes 10 --configs clean_up_both,commons_agents,clean_up_world --out $O ) > $O/alloc.log 2>&1
echo "alloc rc=$?"; tail -52 $O/alloc.log
