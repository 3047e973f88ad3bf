#!/bin/bash
# (GPU box) fp32 build x env groups: with 3072 wave slots (three waves per SIMD) a batch of 4096 / 8192 envs packs better in more, smaller groups
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-fp32g}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
OUT=$O/fp32_groups_ab.txt; : > $OUT
for cfg in 1 2; do
  echo "== configs[$cfg]" >> $OUT
  for lib in ${FP32_LIBS:-libdtrl_f32_w3def libdtrl_f32_w3ilp}; do for g in ${GROUPS_LIST:-2 3 4 6 8}; do
    echo -n "$lib DTRL_GROUPS=$g: " >> $OUT
    DTRL_GROUPS=$g python bench.py --config $cfg --no-cpu-baseline --no-trained-leg --exchange-steps 0 --no-rccl-leg --lib-f32 deepterrainrl_amd/lib/$lib.so 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=d['fp32_physics']
print('fp64 %.3f M | fp32 %.3f M env-steps/s  %.3f ms/frame  kernel avg %.3f ms' % (d['value']/1e6, f['env_steps_per_s']/1e6, f['ms_per_step'], f['kernel_avg_ms']))" >> $OUT 2>&1
  done; done
done
cat $OUT
