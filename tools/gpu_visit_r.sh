#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for v in d24 d20 d16; do python tools/ab/run_ab.py $R/tools/ab/libdtrl_$v.so 1 2; done
done
