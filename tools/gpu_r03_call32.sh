#!/bin/bash
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03r; cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "utcl|tlb|translat|xnack|TCC_EA0?_WRREQ_STALL|TCC_EA0?_WR_UNCACHED|TCC_TAG_STALL|TCC_BUBBLE|WRREQ_DRAM|WRREQ_GMI|WRREQ_IO" | head -80 > $GRAFT_REPO_ROOT/gpurun_out/r03r/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/r03r/counters.txt; head -80 $GRAFT_REPO_ROOT/gpurun_out/r03r/counters.txt | cut -c1-200
