#!/usr/bin/env python3
"""Soak: many frames of a bench workload; every env must stay finite, counters must be monotone. Run via gpurun.
   python tools/soak.py [envs] [frames] [config 1|2] [terrain_gen host|device] [precision f64|f32]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import deepterrainrl_amd as da, bench
n, frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 3000
cfg = bench.CONFIGS[int(sys.argv[3]) if len(sys.argv) > 3 else 1]
tgen = sys.argv[4] if len(sys.argv) > 4 else "host"
prec = sys.argv[5] if len(sys.argv) > 5 else "f64"
b = da.BatchScenario(cfg["arg_file"], n, data_root=bench.ROOT, extra_args={"terrain_seed": 99, "rand_seed": 3, "terrain_gen": tgen, "physics_precision": prec})
b.SetPolicy(bench.xavier_weights(b.PolicyNumParams(), cfg["n_char"], cfg["frag"]), *bench.load_scale(cfg))
prev = b.EvalStats(); t0 = time.time()
for k in range(frames // 250):
    b.RunFrames(250)
    q, qd = b.PoseVel()
    st = b.EvalStats()
    cnt, ids, lam = b.ContactCache()
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(qd)) and np.all(np.isfinite(lam)), "non-finite state"
    assert np.abs(q[:, 1]).max() < 50 and np.abs(qd).max() < 1e4, (np.abs(q[:, 1]).max(), np.abs(qd).max())
    assert cnt.min() >= 0 and cnt.max() <= 24 and np.abs(lam).max() < 1e3, (cnt.min(), cnt.max(), np.abs(lam).max())
    assert st["episodes"] >= prev["episodes"] and st["cycles"] > prev["cycles"]
    prev = st
    print("frame %5d: %s  max|qd| %.1f  max|lambda| %.3f  root x range [%.1f, %.1f]" % ((k + 1) * 250, st, np.abs(qd).max(), np.abs(lam).max(), q[:, 0].min(), q[:, 0].max()), flush=True)
dt = time.time() - t0
print("soak ok (%s, terrain_gen %s, %s): %d envs x %d frames in %.1f s = %.2f M env-steps/s" % (cfg["name"], tgen, prec, n, frames, dt, n * frames * 20 / dt / 1e6))
