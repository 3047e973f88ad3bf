#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_g
mkdir -p $O
cd $R
python bench.py --config 1 --steps 60 --warmup 20 --no-cpu-baseline > $O/bench1.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/bench1.json').read().strip().splitlines()[-1]); print('cfg1', d['value']/1e6, d['roofline']['kernel_avg_ms'], 'exchange', d['exchange']['env_steps_per_s']/1e6, d['exchange']['collective'], 'alt', d['exchange_alt']['env_steps_per_s']/1e6)"
python bench.py --config 2 --steps 60 --warmup 20 --no-cpu-baseline > $O/bench2.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/bench2.json').read().strip().splitlines()[-1]); print('cfg2', d['value']/1e6, d['roofline']['kernel_avg_ms'])"
python tools/env_time_hist.py 8192 120 2 > $O/env_time_hist_cfg2.txt 2>&1; cat $O/env_time_hist_cfg2.txt
python tools/gpu_sections.py 8192 100 2 > $O/sections_cfg2.txt 2>&1; tail -42 $O/sections_cfg2.txt
python tools/env_time_hist.py 4096 120 1 > $O/env_time_hist_cfg1.txt 2>&1; head -3 $O/env_time_hist_cfg1.txt
python tools/train_mace.py --arg-file args/opt_args_train_mace.txt --envs 4096 --frames 400 --trainer hip --overlap 2>&1 | tail -2
