#!/bin/bash
# (GPU box) The opt-in fp32 build: tests, then the bench side figure for each register budget / scheduler variant of its frame kernel -> profiles/r06_fp32_physics.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-fp32}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
[ -n "$SKIP_TESTS" ] || python -m pytest tests/test_fp32_mode.py -m gpu -q -s > $O/pytest_fp32.log 2>&1; [ -n "$SKIP_TESTS" ] || grep -E "passed|failed|HIP fp32|fp32 \{" $O/pytest_fp32.log | cut -c1-420
OUT=$O/fp32_ab.txt; : > $OUT
for cfg in 1 2; do
  echo "== configs[$cfg]" >> $OUT
  for lib in ${FP32_LIBS:-libdtrl_f32}; do
    echo -n "$lib: " >> $OUT
    python bench.py --config $cfg --no-cpu-baseline --no-trained-leg --exchange-steps 0 --no-rccl-leg --lib-f32 deepterrainrl_amd/lib/$lib.so 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); f=d['fp32_physics']
print('fp64 %.3f M env-steps/s (kernel %.3f ms) | fp32 %.3f M env-steps/s  %.3f ms/frame  kernel avg %.3f ms  falls/1000 %.3f vs fp64 %.3f' % (d['value']/1e6, d['roofline']['kernel_avg_ms'], f['env_steps_per_s']/1e6, f['ms_per_step'], f['kernel_avg_ms'], f['falls_per_1000_env_steps'], 1000.0*d['timed_window']['resets']/(d['timed_window']['frames']*20.0*d['config']['envs_per_gpu'])))" >> $OUT 2>&1
  done
done
cat $OUT
