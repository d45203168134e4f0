#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03r; mkdir -p $O
timeout 200 tools/ubench/store_placement > $O/store_placement.md 2>&1; GROUPS=256 timeout 200 tools/ubench/store_placement >> $O/store_placement.md 2>&1; cat $O/store_placement.md
