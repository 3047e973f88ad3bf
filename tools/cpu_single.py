#!/usr/bin/env python3
"""One-off: oracle restatement single-thread rate (SURVEY 8d CPU baseline (i)) for configs 1-3. Run via gpurun (host cores of the GPU box)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import model as om
import bench
ROOT = bench.ROOT
for arg, scale, n, frames in (("args/sim_dog_args.txt", None, 1, 60), ("args/dog_slopes_mixed_args.txt", "data/policies/dog/models/dog_mace3_slopes_mixed_model_scale.txt", 4, 40),
                              ("args/raptor_narrow_gaps_args.txt", "data/policies/raptor/models/raptor_mace3_narrow_gaps_model_scale.txt", 4, 40)):
    m, info = om.build_model(arg, ROOT)
    pol = None
    if scale:
        desc = om.parse_deploy_prototxt(os.path.join(ROOT, info["args"]["policy_net"]))
        io, isc, oo, osc = om.load_scale_file(os.path.join(ROOT, scale))
        pol = (desc, om.xavier_weights(desc, 1234), io, isc, oo, osc)
    t = time.time()
    rate, resets, cycles = om.batch_run(m, n, 1, frames, terrain_seed0=0, rng_seed=0, policy=pol)
    print("%-36s 1 thread, %d envs x %d frames: %.0f env-steps/s (%.1f s)" % (arg, n, frames, rate, time.time() - t), flush=True)
