#!/bin/bash
# (GPU box) launch-order work estimate A/B (VERDICT r5 #5a): shipped row-sum estimate vs last-substep rows vs their mean -> profiles/r06_launch_order_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-order}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
OUT=$O/launch_order_ab.txt; : > $OUT
for cfg in 1 2; do
  echo "== configs[$cfg]" >> $OUT
  for rep in 1 2 3; do for lib in libdtrl libdtrl_cost1 libdtrl_cost2; do
    echo -n "$lib rep $rep: " >> $OUT
    python bench.py --config $cfg --no-cpu-baseline --no-trained-leg --no-fp32-leg --exchange-steps 0 --no-rccl-leg --lib deepterrainrl_amd/lib/$lib.so 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('%.3f M env-steps/s (min %.3f max %.3f)  %.3f ms/frame  kernel avg %.3f ms' % (d['value']/1e6, d['value_min']/1e6, d['value_max']/1e6, d['ms_per_step'], d['roofline']['kernel_avg_ms']))" >> $OUT 2>&1
  done; done
done
cat $OUT
