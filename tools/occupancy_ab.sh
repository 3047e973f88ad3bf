#!/bin/bash
# (GPU box) Pricing of the occupancy lever on the frame kernel -- VERDICT r5 #3 -> profiles/r06_occupancy_ab.txt.  gpurun -- 'bash tools/occupancy_ab.sh <tag>'
#  (1) throughput vs workgroups per CU of the SHIPPED kernel (256 VGPRs, 20 432 B of LDS: 8 per CU = 2 waves per SIMD): DTRL_LDS_PAD adds dynamic LDS -> 7, 6, 5, 4 per CU;
#  (2) the register diet a third wave per SIMD needs, at UNCHANGED occupancy: lib/libdtrl_dyn3.so (workspace in dynamic LDS so that the compiler accepts the budget:
#      168 VGPRs, 106 scratch instructions inside the substep loop) against its control lib/libdtrl_dyn2.so (dynamic LDS, 256 VGPRs) and the shipped library;
#  (3) bitwise equality of the three builds' trajectories (60 frames, 512 envs).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-occ}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
OUT=$O/occupancy_ab.txt; : > $OUT
run() {  # label, env assignments..., then bench args
  label=$1; shift
  for rep in 1 2; do
    echo -n "$label rep $rep: " >> $OUT
    env "$@" python bench.py --no-cpu-baseline --no-trained-leg --exchange-steps 0 --no-rccl-leg $BARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M env-steps/s  %.3f ms/frame  kernel avg %.3f ms  resets/frame %.1f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['timed_window']['resets_per_frame']))" >> $OUT
  done
}
for cfg in 1 2; do
  echo "== configs[$cfg]" >> $OUT
  BARGS="--config $cfg"
  run "shipped, 8 workgroups/CU (2 waves/SIMD)" DTRL_LDS_PAD=0
  run "shipped + 2900 B pad -> 7/CU" DTRL_LDS_PAD=2900
  run "shipped + 6800 B pad -> 6/CU" DTRL_LDS_PAD=6800
  run "shipped + 12200 B pad -> 5/CU" DTRL_LDS_PAD=12200
  run "shipped + 20400 B pad -> 4/CU (1 wave/SIMD)" DTRL_LDS_PAD=20400
  BARGS="--config $cfg --lib deepterrainrl_amd/lib/libdtrl_dyn2.so"; run "dyn2: workspace in dynamic LDS, 256 VGPRs, 8/CU" DTRL_LDS_PAD=0
  BARGS="--config $cfg --lib deepterrainrl_amd/lib/libdtrl_dyn3.so"; run "dyn3: dynamic LDS, 168 VGPRs (budget of 3 waves/SIMD), 8/CU" DTRL_LDS_PAD=0
done
python - >> $OUT <<'PY'
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.getcwd())
import deepterrainrl_amd as da
sys.path.insert(0, "tests")
from conftest import REFDATA, dog_policy
from oracle import model as om
pol = dog_policy(om)
h = {}
for name in ("libdtrl.so", "libdtrl_dyn2.so", "libdtrl_dyn3.so"):
    da.LIB_PATH = os.path.join(os.getcwd(), "deepterrainrl_amd", "lib", name)
    b = da.BatchScenario("args/dog_slopes_mixed_args.txt", 512, data_root=REFDATA, extra_args={"terrain_seed": 3})
    b.SetPolicy(pol[1], *pol[2:]); b.RunFrames(60)
    q, qd = b.PoseVel(); h[name] = hashlib.sha256(q.tobytes() + qd.tobytes()).hexdigest()[:16]; b.close()
print("== trajectories after 60 frames x 512 envs (sha256 of q, qd):", h, "-> bitwise equal" if len(set(h.values())) == 1 else "-> DIFFERENT")
PY
cat $OUT
