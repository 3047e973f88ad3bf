#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for g in 2 3 4; do
DTRL_GROUPS=$g python bench.py --config 1 --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G=$g cfg1', round(d['value']/1e6,2), 'exchange', round(d['exchange']['env_steps_per_s']/1e6,2), 'alt', round(d['exchange_alt']['env_steps_per_s']/1e6,2))"
DTRL_GROUPS=$g python tools/train_mace.py --arg-file args/opt_args_train_mace.txt --envs 4096 --frames 600 --trainer hip --overlap 2>&1 | tail -2 | head -1
done
