R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lds_fwd2; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/gpu_sections.py 4096 30 > $O/sections.log 2>&1; grep -i "kernel\|policy forward\|Total\|Action" $O/sections.log
bash tools/gpu_profile.sh r02_v11 > $O/profile.log 2>&1; tail -5 $O/profile.log
