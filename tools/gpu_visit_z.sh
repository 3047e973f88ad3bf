#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2 3; do
for c in 1 2; do
for v in w0 w1; do python tools/ab/run_ab.py $R/tools/ab/libdtrl_$v.so $c 2; done
done
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -a "passed\|failed" | tail -2
