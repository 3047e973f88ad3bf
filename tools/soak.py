#!/usr/bin/env python3
"""Soak: many frames of the bench workload; every env must stay finite, counters must be monotone. Run via gpurun."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import deepterrainrl_amd as da, bench
n, frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 3000
b = da.BatchScenario(bench.CONFIGS[1]["arg_file"], n, data_root=bench.ROOT, extra_args={"terrain_seed": 99, "rand_seed": 3})
b.SetPolicy(bench.xavier_weights(b.PolicyNumParams()), *bench.load_scale(bench.CONFIGS[1]))
prev = b.EvalStats(); t0 = time.time()
for k in range(frames // 250):
    b.RunFrames(250)
    q, qd = b.PoseVel()
    st = b.EvalStats()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(qd)), "non-finite state"
    assert np.abs(q[:, 1]).max() < 50 and np.abs(qd).max() < 1e4, (np.abs(q[:, 1]).max(), np.abs(qd).max())
    assert st["episodes"] >= prev["episodes"] and st["cycles"] > prev["cycles"]
    prev = st
    print("frame %5d: %s  max|qd| %.1f  root x range [%.1f, %.1f]" % ((k + 1) * 250, st, np.abs(qd).max(), q[:, 0].min(), q[:, 0].max()), flush=True)
dt = time.time() - t0
print("soak ok: %d envs x %d frames in %.1f s = %.2f M env-steps/s" % (n, frames, dt, n * frames * 20 / dt / 1e6))
