#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for c in 1 2; do
for g in 2 3 4; do python tools/ab/run_ab.py $R/deepterrainrl_amd/lib/libdtrl.so $c $g; done
done
done
