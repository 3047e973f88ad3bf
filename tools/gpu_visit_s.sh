#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for v in d20 h16_12 h12_12 h20_16 h12_8; do python tools/ab/run_ab.py $R/tools/ab/libdtrl_$v.so 1 2; done
for v in d24 h16_12 h12_12 h20_16 h12_8; do python tools/ab/run_ab.py $R/tools/ab/libdtrl_$v.so 2 2; done
done
