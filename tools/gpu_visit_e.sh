#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for cfg in 1 2; do for g in 2 4; do
  echo "== config $cfg DTRL_GROUPS=$g"; DTRL_GROUPS=$g python bench.py --config $cfg --steps 60 --warmup 20 --no-cpu-baseline --exchange-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['roofline']['kernel_avg_ms'])"
done; done
