#!/bin/bash
# AddressSanitizer + UBSan pass over the host engine and the lane-loop build of the kernel source (CPU only, test backend of tests/emul):
# exploration / raptor / goat / sim scenarios, a prone pose that saturates the row cap, resets, drains, getters, the policy forward on other conv shapes.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/deepterrainrl_amd/csrc
g++ -O1 -g -mfma -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -I. \
    -o /tmp/libdtrl_emul_asan.so dtrl_host.cpp dtrl_engine.cpp dtrl_c_api.cpp $R/tests/emul/dtrl_backend_emul.cpp -lpthread
g++ -O1 -g -mfma -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -I. \
    -o /tmp/libdtrl_trainer_emul_asan.so $R/tests/emul/dtrl_trainer_emul.cpp
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
# round 3: tuple rings in host memory, pipelined drains (plain + packed with a small block: carry-over), dtrl_step_poll, policy hand-over during a frame
os.environ["DTRL_GROUPS"] = "2"
b = AsanScenario("args/opt_args_train_mace.txt", 10, data_root=REFDATA, extra_args={"terrain_seed": 7, "exp_base_rate": 0.3, "tuple_ring": "host"})
p = dog_policy(om); b.SetPolicy(p[1], *p[2:])
w = np.ascontiguousarray(p[1], np.float32)
blk = np.zeros((4 + 1, b.W + 2), np.float32)
b.SetTuplePipelining(True); b.UpdateBegin()
n_rows = 0
for f in range(90):
    b.UpdateEndBegin()
    if f % 3 == 0:
        n_rows += b.DrainTuplesPacked(blk.ctypes.data, 4, want_count=True)
    else:
        n_rows += len(b.DrainTuples()[0])
    if f % 5 == 0:
        b.SetPolicyDevice(w.ctypes.data, w.size)
    b.UpdatePoll()
b.UpdateEndBegin(); b.UpdateEnd()
for _ in range(40):
    if not len(b.DrainTuples()[0]) and b.TupleStats()["pending"] == 0: break
    b.UpdateBegin(); b.UpdateEnd()
print("pipelined host ring ok", n_rows, b.TupleStats())
b.close()
# the native trainer step's plain-loop build: staged tuple stores through a ring wrap, fused critic / actor calls
import torch, test_trainer as TT
from deepterrainrl_amd import hip_trainer as ht
t = ht.HipMACETrainer(TT.TRAIN, TT.SOLVER, TT.S, TT.A, lib_path="/tmp/libdtrl_trainer_emul_asan.so", mem_size=128, num_init_samples=60, freeze_target_iters=2, device="cpu", seed=3)
rows, flags = TT.random_rows(np.random.RandomState(2), 200, p_actor=0.5)
k = 0
while k < 200:
    st = t.StageTuples(rows[k:], flags[k:])
    for j in range(0, st, 32):
        t.AddTuples(rows[k + j:k + j + 32], flags[k + j:k + j + 32], staged=j); t.Train()
    k += st
print("trainer ok", t.GetIter(), t.actor_iter)
PY
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) python /tmp/asan_run.py
