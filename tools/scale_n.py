#!/usr/bin/env python3
"""Throughput vs batch size on one GPU (dog slopes_mixed, policy on). Usage: [TERRAIN_GEN=device] tools/scale_n.py 2048 4096 ... (run via gpurun)"""
import sys, os, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import deepterrainrl_amd as da
import bench
for n in [int(a) for a in sys.argv[1:]]:
    t0 = time.time()
    b = da.BatchScenario(bench.CONFIGS[1]["arg_file"], n, data_root=bench.ROOT, extra_args={"terrain_seed": 1, "rand_seed": 1, "terrain_gen": os.environ.get("TERRAIN_GEN", "host")})
    b.SetPolicy(bench.xavier_weights(b.PolicyNumParams()), *bench.load_scale(bench.CONFIGS[1]))
    t1 = time.time()
    b.RunFrames(20); b.KernelTimeMs()
    b.EvalStats()   # (synchronises: in device mode RunFrames only queues)
    t = time.time(); b.RunFrames(30); b.EvalStats(); dt = time.time() - t
    ms, nl = b.KernelTimeMs()
    print("envs=%6d: create %.1f s, %.2f M env-steps/s wall (kernel %.2f ms/frame, host+launch %.2f ms/frame), stats %s" % (
        n, t1 - t0, n * 30 * 20 / dt / 1e6, ms, dt / 30 * 1e3 - ms, b.EvalStats()), flush=True)
    b.close()
