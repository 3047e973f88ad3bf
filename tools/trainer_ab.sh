#!/bin/bash
# (GPU box) native trainer step: DTRL_TRAINER_FUSED modes side by side + GPU tests under the split forward + a kernel table -> profiles/r06_trainer.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-trainer}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
OUT=$O/trainer_ab.txt; : > $OUT
for mode in ${MODES:-1 3}; do
  echo "== DTRL_TRAINER_FUSED=$mode" >> $OUT
  DTRL_TRAINER_FUSED=$mode python tools/trainer_rate.py --iters 1000 --only hip 2>&1 | grep "Train()" >> $OUT
done
for mode in ${TEST_MODES:-3}; do
  DTRL_TRAINER_FUSED=$mode python -m pytest tests/test_hip_trainer.py tests/test_reference_learn.py -m gpu -q > $O/pytest_trainer_mode$mode.log 2>&1; echo "GPU trainer tests under DTRL_TRAINER_FUSED=$mode: $(tail -1 $O/pytest_trainer_mode$mode.log)" >> $OUT
done
cd /tmp && export TMPDIR=/tmp
for mode in ${PROF_MODES:-3}; do
  DTRL_TRAINER_FUSED=$mode rocprofv3 --kernel-trace --stats -d $O/stats_mode$mode -o stats -- python $R/tools/trainer_rate.py --iters 200 --repeats 1 --only hip > $O/stats_mode$mode.log 2>&1
  echo "== kernel table, DTRL_TRAINER_FUSED=$mode (rocprofv3 --kernel-trace --stats -- python tools/trainer_rate.py --iters 200 --repeats 1 --only hip)" >> $OUT
  python $R/tools/rocpd_top.py $(find $O/stats_mode$mode -name "*.db" | head -1) 16 >> $OUT 2>&1
  find $O/stats_mode$mode -name "*.db" -delete
done
cat $OUT
