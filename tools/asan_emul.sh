#!/bin/bash
# AddressSanitizer + UBSan pass over the host engine and the lane-loop build of the kernel source (CPU only, test backend of tests/emul):
# exploration / raptor / goat / sim scenarios, a prone pose that saturates the row cap, resets, drains, getters, the policy forward on other conv shapes.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/deepterrainrl_amd/csrc
g++ -O1 -g -mfma -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -I. \
    -o /tmp/libdtrl_emul_asan.so dtrl_host.cpp dtrl_engine.cpp dtrl_c_api.cpp $R/tests/emul/dtrl_backend_emul.cpp -lpthread
cat > /tmp/asan_run.py <<PY
import sys, os
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import numpy as np
from conftest import REFDATA, dog_policy
from oracle import model as om
import deepterrainrl_amd as da
import test_host_and_emul as T
class AsanScenario(da.BatchScenario):
    def _library(self):
        return da._bind("/tmp/libdtrl_emul_asan.so")
for arg, n, frames in (("args/opt_args_train_mace.txt", 9, 120), ("args/raptor_narrow_gaps_args.txt", 5, 80), ("args/goat_cliffs_args.txt", 5, 60), ("args/sim_dog_args.txt", 3, 40)):
    b = AsanScenario(arg, n, data_root=REFDATA, extra_args={"terrain_seed": 5, "exp_base_rate": 0.3})
    if b.PolicyNumParams():
        p = T.raptor_policy(om) if "raptor" in arg else dog_policy(om)
        b.SetPolicy(p[1], *p[2:])
    b.RunFrames(frames)
    q, qd = b.PoseVel(); q[0][2:] = 0; q[0][1] = 0.06 + b.SampleGround(0, [q[0][0]])[0][0]; b.SetPoseVel(q, qd); b.RunFrames(5)
    b.Reset([0]); b.DrainTuples(); b.RecordPoliState(); b.Contacts(); b.Ctrl()
    if b.PolicyNumParams(): b.PolicyOutput()
    print(arg, "ok", b.EvalStats())
    b.close()
PY
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) python /tmp/asan_run.py
