#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_hip_trainer.py -m gpu -q -x -k "q_and_cacla or compiled_reference" 2>&1 | grep -a "passed\|failed" | tail -2
for a in args/opt_args_train_q.txt args/opt_args_train_cacla.txt; do
  python tools/train_mace.py --arg-file $a --envs 4096 --frames 300 --trainer hip --overlap --init-samples 5000 2>&1 | grep -a "env-steps/s" | tail -1
done
python tools/train_mace.py --arg-file args/opt_args_train_mace.txt --envs 4096 --frames 600 --trainer hip --overlap 2>&1 | grep -a "env-steps/s" | tail -1
python bench.py --config 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330
