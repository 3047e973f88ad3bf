#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_trainer.py tests/test_trainer.py -m gpu -q -x > $O/pytest_trainer.log 2>&1; grep -E "passed|failed|Error" $O/pytest_trainer.log | tail -5
echo "== fused"; python tools/trainer_rate.py --iters 1000 --only hip 2>&1 | grep -i "train" | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/tr_stats -o tr -- python $R/tools/trainer_rate.py --iters 200 --repeats 1 --only hip > $O/tr_stats.log 2>&1
DB=$(find $O/tr_stats -name "*.db" | head -1); python $R/tools/rocpd_top.py $DB 12 > $O/trainer_top.txt 2>&1; find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
head -10 $O/trainer_top.txt
