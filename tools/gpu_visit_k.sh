#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2 3; do
for c in 1 2; do
for v in k0 k1; do python tools/ab/run_ab.py $R/tools/ab/libdtrl_$v.so $c 2; done
done
done
