#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/tr_stats -o tr -- python $R/tools/trainer_rate.py --iters 200 --repeats 1 --only hip > $O/tr_stats.log 2>&1
DB=$(find $O/tr_stats -name "*.db" | head -1); python $R/tools/rocpd_top.py $DB 30 > $O/trainer_top.txt 2>&1; find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
cat $O/trainer_top.txt
