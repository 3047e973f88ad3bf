#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r04_n
python -m pytest tests -m gpu -q -x > gpurun_out/r04_n/pytest_gpu.log 2>&1; tail -25 gpurun_out/r04_n/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --config 1 2>/dev/null | tail -1 > gpurun_out/r04_n/bench1.json; cut -c1-400 gpurun_out/r04_n/bench1.json
python bench.py --config 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_n/bench2.json; cut -c1-400 gpurun_out/r04_n/bench2.json
