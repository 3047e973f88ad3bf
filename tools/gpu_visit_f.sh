#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do for cfg in 1 2; do for n in 12 18 24; do for g in 2 4; do
  python tools/ab/run_ab.py $R/tools/ab/libdtrl_$n.so $cfg $g
done; done; done; done
