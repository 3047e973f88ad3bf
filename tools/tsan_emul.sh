#!/bin/bash
# ThreadSanitizer pass over the host engine (CPU only, lane-loop backend): the frame-boundary worker pool (terrain regeneration across threads) under resets.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/deepterrainrl_amd/csrc
g++ -O1 -g -mfma -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=thread -fno-omit-frame-pointer -I. \
    -o /tmp/libdtrl_emul_tsan.so dtrl_host.cpp dtrl_engine.cpp dtrl_c_api.cpp $R/tests/emul/dtrl_backend_emul.cpp -lpthread
cat > /tmp/tsan_run.py <<PY
import sys, os
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
os.environ["DTRL_HOST_THREADS"] = "6"
from conftest import REFDATA, dog_policy
from oracle import model as om
import deepterrainrl_amd as da
class TsanScenario(da.BatchScenario):
    def _library(self):
        return da._bind("/tmp/libdtrl_emul_tsan.so")
for arg, n, frames in (("args/goat_cliffs_args.txt", 24, 40), ("args/opt_args_train_mace.txt", 24, 40)):
    b = TsanScenario(arg, n, data_root=REFDATA, extra_args={"terrain_seed": 5, "exp_base_rate": 0.3})
    p = dog_policy(om); b.SetPolicy(p[1], *p[2:])
    b.RunFrames(frames)
    for f in range(10):
        b.Update(); b.DrainTuples()
    print(arg, "ok", b.EvalStats(), flush=True)
    b.close()
PY
TSAN_OPTIONS="report_signal_unsafe=0 halt_on_error=0" LD_PRELOAD=$(g++ -print-file-name=libtsan.so) python /tmp/tsan_run.py
