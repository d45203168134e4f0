#!/bin/bash
# dev helper: memory-side PMC counters for bench.py (env MP_RENDER_ABLATE honoured) and the store micro-benchmark
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_mem; mkdir -p $O; cd /tmp
TCP="TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum"
TCC="TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum"
for a in 0 10; do
  MP_RENDER_ABLATE=$a timeout -k 5 90 rocprofv3 --pmc $TCP -d $O/tcp_$a -o r -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/tcp_$a.log 2>&1
  MP_RENDER_ABLATE=$a timeout -k 5 90 rocprofv3 --pmc $TCC -d $O/tcc_$a -o r -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/tcc_$a.log 2>&1
done
timeout -k 5 60 rocprofv3 --pmc $TCP -d $O/tcp_ub -o r -- $R/tools/ubench/store_bw2 > $O/tcp_ub.log 2>&1
timeout -k 5 60 rocprofv3 --pmc $TCC -d $O/tcc_ub -o r -- $R/tools/ubench/store_bw2 > $O/tcc_ub.log 2>&1
ls $O
