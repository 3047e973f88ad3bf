#!/bin/bash
# GPU-box visit A (round 4): the -m gpu suite, smoke(), both bench configs, the per-section cycle breakdown.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -rP > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --config 1 --steps 60 --warmup 20 > $O/bench1.json 2> $O/bench1.err; tail -c 600 $O/bench1.json
python bench.py --config 2 --steps 60 --warmup 20 --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err; tail -c 600 $O/bench2.json
