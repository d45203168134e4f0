#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O; export PYTHONUNBUFFERED=1 PYTHONPATH=$R
cd $R
run() { # lib view substrate worlds
  [ -n "$1" ] && export MP_ENGINE_LIB=$R/meltingpot_amd/lib/libmp_engine_$1.so || unset MP_ENGINE_LIB
  echo "== lib [$1] $3 $2: $(VIEW=$2 timeout 120 python tools/gpu_bimodal3.py $3 $4 many_buffers 2>&1 | grep many | sed 's/.*step //; s/ us//' | tr '\n' ' ')"
}
for lib in sc1_t4 sc1_t5 sc1_t6 sc1nt_t0 sc1nt_t4 sc01_t4 sc0_t4; do run "$lib" agents commons_harvest__open 4096; done > $O/policy_tokens2.txt 2>&1
for lib in "" sc1_t0 sc1_t4 sc1_t6 sc1nt_t4; do run "$lib" agents territory__rooms 8192; done >> $O/policy_tokens2.txt 2>&1
for lib in "" sc1_t4 sc1_t6; do run "$lib" agents prisoners_dilemma_in_the_matrix__arena 8192; done >> $O/policy_tokens2.txt 2>&1
for lib in sc1_t6 sc1nt_t4 sc0_t4; do run "$lib" world clean_up 4096; done >> $O/policy_tokens2.txt 2>&1
cat $O/policy_tokens2.txt
