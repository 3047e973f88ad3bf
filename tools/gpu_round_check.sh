#!/bin/bash
# One GPU-box visit at HEAD: the -m gpu suite, smoke(), the two bench configs' profiles, the trainer's rate + kernel table, the training loops.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_head
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/gpu_profile.sh r04_cfg1 1 > $O/prof1.log 2>&1
bash tools/gpu_profile.sh r04_cfg2 2 > $O/prof2.log 2>&1
python tools/trainer_rate.py --iters 1000 > $O/trainer_rate.log 2>&1; cat $O/trainer_rate.log | grep Train
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/tr_stats -o tr -- python $R/tools/trainer_rate.py --iters 100 --repeats 1 --only hip > $O/tr_stats.log 2>&1
DB=$(find $O/tr_stats -name "*.db" | head -1); python $R/tools/rocpd_top.py $DB 30 > $O/trainer_top.txt 2>&1; find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
cd $R
for spec in "args/opt_args_train_mace.txt 4096" "args/opt_args_train_goat_mace.txt 8192" "args/opt_args_train_raptor_mace.txt 8192"; do
  set -- $spec
  echo "== $1 envs=$2 frames=600 trainer=hip --overlap" >> $O/train_loops.log
  python tools/train_mace.py --arg-file $1 --envs $2 --frames 600 --trainer hip --overlap 2>&1 | tail -2 >> $O/train_loops.log
done
for a in args/opt_args_train_q.txt args/opt_args_train_cacla.txt; do
  echo "== $a envs=4096 frames=300 trainer=hip --overlap --init-samples 5000" >> $O/train_loops.log
  python tools/train_mace.py --arg-file $a --envs 4096 --frames 300 --trainer hip --overlap --init-samples 5000 2>&1 | grep -a "env-steps/s" | tail -1 >> $O/train_loops.log
done
echo "== args/opt_args_train_mace.txt envs=4096 frames=600 trainer=hip (sequential)" >> $O/train_loops.log
python tools/train_mace.py --arg-file args/opt_args_train_mace.txt --envs 4096 --frames 600 --trainer hip 2>&1 | tail -2 >> $O/train_loops.log
cat $O/train_loops.log
# the exchange through a one-rank RCCL group and the training loop across ranks (one rank)
DTRL_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_forced_rccl.json
python tools/train_mace.py --distributed --overlap --envs 4096 --frames 600 --trainer hip 2>&1 | grep "distributed" > $O/train_distributed.log; cat $O/train_distributed.log
# per-section cycles of the frame kernel (profile build) for both configs, and a short soak
cd $R
python tools/gpu_sections.py 4096 60 1 > $O/sections_cfg1.txt 2>&1
python tools/gpu_sections.py 8192 60 2 > $O/sections_cfg2.txt 2>&1
python tools/soak.py 4096 1500 2>&1 | tail -2 > $O/soak.txt; cat $O/soak.txt
