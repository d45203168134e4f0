#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout 120 python $R/tools/gpu_placement.py commons_harvest__open 4096 prof_buffers 2>&1 | grep prof > $O/prof_plain.txt
timeout 300 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY -d /tmp/pa -o pa --output-format csv -- python $R/tools/gpu_placement.py commons_harvest__open 4096 prof_buffers 2>&1 | grep prof > $O/prof_a.txt
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_TAG_STALL_sum -d /tmp/pb -o pb --output-format csv -- python $R/tools/gpu_placement.py commons_harvest__open 4096 prof_buffers 2>&1 | grep prof > $O/prof_b.txt
for d in pa pb; do for f in $(find /tmp/$d -name "*.csv"); do cp $f $O/${d}_$(basename $f); done; done
ls -la $O | tail -12; cat $O/prof_plain.txt $O/prof_a.txt $O/prof_b.txt
