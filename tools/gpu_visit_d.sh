#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_d
mkdir -p $O
cd $R
for cfg in 1 2; do for g in 2 4; do
  echo "== config $cfg DTRL_GROUPS=$g"; DTRL_GROUPS=$g python bench.py --config $cfg --steps 60 --warmup 20 --no-cpu-baseline --exchange-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['roofline']['kernel_avg_ms'])"
done; done | tee $O/groups.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bitwise or oracle or 1200" 2>&1 | tail -3
