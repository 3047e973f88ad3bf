#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r04_final
python -m pytest tests -m gpu -q -x > gpurun_out/r04_final/pytest_gpu.log 2>&1; grep -a "passed\|failed" gpurun_out/r04_final/pytest_gpu.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_profile.sh r04_cfg1 1 > gpurun_out/prof1.log 2>&1
bash tools/gpu_profile.sh r04_cfg2 2 > gpurun_out/prof2.log 2>&1
tail -c 400 gpurun_out/r04_cfg1/bench.json
