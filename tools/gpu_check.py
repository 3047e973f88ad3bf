#!/usr/bin/env python3
"""Scratch GPU check: HIP path vs oracle on a few envs, then a short throughput probe. Run via gpurun."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import model as om
import deepterrainrl_amd as da
ROOT = os.path.join(REPO, "tests", "golden", "refdata")

def policy(desc_path, scale_path):
    desc = om.parse_deploy_prototxt(desc_path)
    w = om.xavier_weights(desc, 1234)
    io, isc, oo, osc = om.load_scale_file(scale_path)
    return desc, w, io, isc, oo, osc

def parity(arg_file, n_envs, n_steps, use_policy):
    m, info = om.build_model(arg_file, ROOT)
    pol = None
    if use_policy:
        pol = policy(os.path.join(ROOT, info["args"]["policy_net"]), os.path.join(ROOT, "data/policies/dog/models/dog_mace3_slopes_mixed_model_scale.txt"))
    b = da.BatchScenario(arg_file, n_envs, data_root=ROOT, extra_args={"terrain_seed": 100})
    if pol: b.SetPolicy(pol[1], *pol[2:])
    es = [om.OracleEnv(m, terrain_seed=100 + i, rng_seed=0, env_id=i, policy=pol) for i in range(n_envs)]
    worst = 0
    for k in range(n_steps // 20):
        b.StepUpdates(20)
        for e in es: e.step(20)
        q, qd = b.PoseVel()
        d = max(max(abs(q[i] - es[i].pose_vel()[0]).max(), abs(qd[i] - es[i].pose_vel()[1]).max()) for i in range(n_envs))
        worst = max(worst, d)
    print("parity %s envs=%d steps=%d policy=%s: max |dq|,|dqd| = %.3e" % (arg_file, n_envs, n_steps, use_policy, worst), flush=True)

def throughput(arg_file, n_envs, frames, use_policy):
    m, info = om.build_model(arg_file, ROOT)
    b = da.BatchScenario(arg_file, n_envs, data_root=ROOT, extra_args={"terrain_seed": 1})
    if use_policy:
        pol = policy(os.path.join(ROOT, info["args"]["policy_net"]), os.path.join(ROOT, "data/policies/dog/models/dog_mace3_slopes_mixed_model_scale.txt"))
        b.SetPolicy(pol[1], *pol[2:])
    b.RunFrames(3)
    b.KernelTimeMs()
    t = time.time(); b.RunFrames(frames); dt = time.time() - t
    kms, nl = b.KernelTimeMs()
    print("throughput %s envs=%d: %.0f env-steps/s wall, kernel %.3f ms/frame (%d launches) -> %.0f env-steps/s kernel-only; stats %s" % (
        arg_file, n_envs, n_envs * frames * 20 / dt, kms, nl, n_envs * 20 / (kms * 1e-3), b.EvalStats()), flush=True)

if __name__ == "__main__":
    print(da.version())
    parity("args/sim_dog_args.txt", 2, 240, False)
    parity("args/dog_slopes_mixed_args.txt", 4, 600, True)
    throughput("args/sim_dog_args.txt", 4096, 20, False)
    throughput("args/dog_slopes_mixed_args.txt", 4096, 20, True)
