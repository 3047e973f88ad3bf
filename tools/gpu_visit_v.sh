#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r04_v
for rep in 1 2; do
python tools/trainer_rate.py --iters 1000 --only hip 2>&1 | grep -a "Train()/s" | tail -1
for v in t1 t2; do python tools/trainer_rate.py --iters 1000 --only hip --lib $R/tools/ab/libdtrl_$v.so 2>&1 | grep -a "Train()/s" | tail -1; done
done
python -m pytest tests -m gpu -q -x > gpurun_out/r04_v/pytest_gpu.log 2>&1; grep -a "passed\|failed" gpurun_out/r04_v/pytest_gpu.log | tail -2
python bench.py --config 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
python bench.py --config 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
