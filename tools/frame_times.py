#!/usr/bin/env python3
"""Per-frame kernel time of the bench workload (which frames are slow, and why). Run via gpurun."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import deepterrainrl_amd as da
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = da.BatchScenario(bench.CONFIGS[1]["arg_file"], n, data_root=bench.ROOT, extra_args={"terrain_seed": 20260925, "rand_seed": 1})
b.SetPolicy(bench.xavier_weights(b.PolicyNumParams()), *bench.load_scale(bench.CONFIGS[1]))
prev = b.EvalStats()
rows = []
for f in range(100):
    b.KernelTimeMs()
    b.RunFrames(1)
    ms, nl = b.KernelTimeMs()
    st = b.EvalStats()
    fl = b.Flags()
    rows.append((f, ms, st["cycles"] - prev["cycles"], st["resets"] - prev["resets"], int(np.sum((fl & 1) != 0)), int(np.sum((fl & 2) != 0))))
    prev = st
for r in rows:
    print("frame %3d: kernel %7.3f ms  new cycles %4d  resets %3d  fallen %4d stumbled %4d" % r)
