#!/bin/bash
# GPU-box visit B (round 4): the whole -m gpu suite (no -x), per-section cycle breakdown, full profile of config 1 (stats + PMC passes incl. lane occupancy)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_b
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -rP > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python tools/gpu_sections.py 4096 100 > $O/sections.txt 2>&1; tail -45 $O/sections.txt
bash tools/gpu_profile.sh r04_cfg1 1 > $O/prof1.log 2>&1; tail -30 $R/gpurun_out/r04_cfg1/summary.txt
