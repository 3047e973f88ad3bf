#!/bin/bash
# GPU visit C: per-env time distribution, group-count sweep, device-terrain headline
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_c
mkdir -p $O
cd $R
python tools/env_time_hist.py 4096 120 1 > $O/env_time_hist.txt 2>&1; cat $O/env_time_hist.txt
for g in 2 3 4; do
  echo "== DTRL_GROUPS=$g"; DTRL_GROUPS=$g python bench.py --config 1 --steps 60 --warmup 20 --no-cpu-baseline --exchange-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['roofline']['kernel_avg_ms'])"
done | tee $O/groups.txt
echo "== terrain-gen device"; python bench.py --config 1 --steps 60 --warmup 20 --no-cpu-baseline --exchange-steps 0 --terrain-gen device 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['roofline']['kernel_avg_ms'])" | tee $O/device_terrain.txt
