#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03q; mkdir -p $O
INTERLEAVE=1 timeout 200 tools/ubench/store_ceiling > $O/store_interleave.md 2>&1; tail -30 $O/store_interleave.md
