#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for t in 8 2 1; do echo "DTRL_HOST_THREADS=$t"; DTRL_HOST_THREADS=$t python tools/ab/run_ab.py $R/deepterrainrl_amd/lib/libdtrl.so 1 2; DTRL_HOST_THREADS=$t python tools/ab/run_ab.py $R/deepterrainrl_amd/lib/libdtrl.so 2 2; done
