#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_i
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -rP > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
python bench.py --config 1 --steps 60 --warmup 20 > $O/bench1.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/bench1.json').read().strip().splitlines()[-1]); print('cfg1', d['value']/1e6, d['roofline']['kernel_avg_ms'], 'exchange', d['exchange']['env_steps_per_s']/1e6, 'alt', d['exchange_alt']['env_steps_per_s']/1e6)"
python bench.py --config 2 --steps 60 --warmup 20 --no-cpu-baseline > $O/bench2.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/bench2.json').read().strip().splitlines()[-1]); print('cfg2', d['value']/1e6, d['roofline']['kernel_avg_ms'])"
