#!/usr/bin/env python3
"""Where one frame of bench.py's exchange leg spends its host time (run via gpurun)."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
import deepterrainrl_amd as da
from deepterrainrl_amd.sharding import ShardedRollout
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
make = lambda nl, off: da.BatchScenario(bench.EXCHANGE_ARG_FILE, nl, data_root=bench.ROOT, extra_args={"terrain_seed": 20260925, "rand_seed": 1, "global_env_offset": off})
sr = ShardedRollout(make, n, device=dev)
b = sr.batch
sr.broadcast_policy(bench.xavier_weights(b.PolicyNumParams()), *bench.load_scale(bench.CONFIGS[1]))
b.SetExplore(True, 0.2, 0.025, 0.002)
T = dict(end=0.0, gbegin=0.0, begin=0.0, gend=0.0, append=0.0)
replay = torch.zeros((1 << 18, b.W), device=dev); cur = 0
sr.UpdateBegin()
def frame(rec):
    global cur
    t0 = time.perf_counter(); sr.UpdateEnd()
    t1 = time.perf_counter()
    g = sr.gather_tuples_end() if sr._pending is not None else None
    t2 = time.perf_counter()
    if g is not None:
        m = int(g[0].shape[0])
        if m:
            idx = (torch.arange(m, device=dev) + cur) % replay.shape[0]; replay[idx] = g[0]; cur += m
    t3 = time.perf_counter(); sr.gather_tuples_begin()
    t4 = time.perf_counter(); sr.UpdateBegin()
    t5 = time.perf_counter()
    if rec:
        for k, v in zip(("end", "gend", "append", "gbegin", "begin"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[k] += v
for k in range(20): frame(False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(50): frame(True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("ms per frame %.3f; phases (ms): %s" % (dt / 50 * 1e3, {k: round(v / 50 * 1e3, 3) for k, v in T.items()}))
st = b.EvalStats(); print(st, b.TupleStats())
sr.UpdateEnd()
if sr._pending is not None: sr.gather_tuples_end()
# same workload through RunFrames (no exchange) for reference
b.RunFrames(20); torch.cuda.synchronize(); t0 = time.perf_counter(); b.RunFrames(50); torch.cuda.synchronize()
print("RunFrames only: %.3f ms per frame" % ((time.perf_counter() - t0) / 50 * 1e3))
t0 = time.perf_counter()
for k in range(50): b.Update()
torch.cuda.synchronize(); print("Update loop only: %.3f ms per frame" % ((time.perf_counter() - t0) / 50 * 1e3))
