#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
bash tools/gpu_profile.sh r04_cfg1 1 > gpurun_out/prof1.log 2>&1
bash tools/gpu_profile.sh r04_cfg2 2 > gpurun_out/prof2.log 2>&1
python tools/train_mace.py --arg-file args/opt_args_train_mace.txt --envs 4096 --frames 600 --trainer hip --overlap 2>&1 | grep -a "env-steps/s" | tail -1
python tools/train_mace.py --arg-file args/opt_args_train_raptor_mace.txt --envs 8192 --frames 600 --trainer hip --overlap 2>&1 | grep -a "env-steps/s" | tail -1
python tools/train_mace.py --arg-file args/opt_args_train_goat_mace.txt --envs 8192 --frames 600 --trainer hip --overlap 2>&1 | grep -a "env-steps/s" | tail -1
for a in args/opt_args_train_q.txt args/opt_args_train_cacla.txt; do python tools/train_mace.py --arg-file $a --envs 4096 --frames 300 --trainer hip --overlap --init-samples 5000 2>&1 | grep -a "env-steps/s" | tail -1; done
python tools/train_mace.py --distributed --overlap --envs 4096 --frames 600 --trainer hip 2>&1 | grep -a "distributed" | head -1
