#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for v in base nest; do python tools/ab/run_ab.py $R/tools/ab/libdtrl_$v.so 1 2; done
for v in base nest nest_r16 nest_r24; do python tools/ab/run_ab.py $R/tools/ab/libdtrl_$v.so 2 2; done
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
