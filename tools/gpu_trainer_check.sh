#!/bin/bash
# The trainer's numbers at HEAD on one GPU box: rate (native + peer), kernel table under rocprofv3, the three training loops (overlapped) + dog sequential.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_trainer; mkdir -p $O
cd $R
python tools/trainer_rate.py --iters 1000 > $O/trainer_rate.log 2>&1; grep Train $O/trainer_rate.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/tr_stats -o tr -- python $R/tools/trainer_rate.py --iters 100 --repeats 1 --only hip > $O/tr_stats.log 2>&1
DB=$(find $O/tr_stats -name "*.db" | head -1); python $R/tools/rocpd_top.py $DB 24 > $O/trainer_top.txt 2>&1; find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
cd $R
for spec in "args/opt_args_train_mace.txt 4096" "args/opt_args_train_goat_mace.txt 8192" "args/opt_args_train_raptor_mace.txt 8192"; do
  set -- $spec
  echo "== $1 envs=$2 frames=600 trainer=hip --overlap" >> $O/train_loops.log
  python tools/train_mace.py --arg-file $1 --envs $2 --frames 600 --trainer hip --overlap 2>&1 | tail -3 >> $O/train_loops.log
done
echo "== args/opt_args_train_mace.txt envs=4096 frames=600 trainer=hip --overlap (second run)" >> $O/train_loops.log
python tools/train_mace.py --arg-file args/opt_args_train_mace.txt --envs 4096 --frames 600 --trainer hip --overlap 2>&1 | tail -2 >> $O/train_loops.log
echo "== args/opt_args_train_mace.txt envs=4096 frames=600 trainer=hip (sequential)" >> $O/train_loops.log
python tools/train_mace.py --arg-file args/opt_args_train_mace.txt --envs 4096 --frames 600 --trainer hip 2>&1 | tail -2 >> $O/train_loops.log
cat $O/train_loops.log
