#!/usr/bin/env python3
"""Frame launches of the bench workload with or without the policy net, for PMC passes that attribute HBM traffic (run under rocprofv3 --pmc).
Usage: traffic_probe.py net|nonet [envs] [frames]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import deepterrainrl_amd as da
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "net"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 30
b = da.BatchScenario(bench.CONFIGS[1]["arg_file"], n, data_root=bench.ROOT, extra_args={"terrain_seed": 20260925, "rand_seed": 1})
if mode == "net":
    b.SetPolicy(bench.xavier_weights(b.PolicyNumParams()), *bench.load_scale(bench.CONFIGS[1]))
b.RunFrames(frames)
print(mode, b.EvalStats())
