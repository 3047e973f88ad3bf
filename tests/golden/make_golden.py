#!/usr/bin/env python3
"""Freeze golden OUTPUT vectors from the CPU oracle (run here, committed as small .npz files):
  terrain_golden.npz   heights for (terrain type, seed) pairs           (cTerrainGen2D restatement, libstdc++ streams)
  sim_dog_trace.npz    q/qd every 20 env-steps over 240 env-steps (= 1200 substeps) of args/sim_dog_args.txt (BASELINE config 0)
  mace_tuples.npz      first tuples of a 2-env exploration run with the synthetic policy (MACE replay rows + flags)
  nn_golden.npz        policy-net output for a fixed input with the seed-1234 synthetic weights
Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import model as om  # noqa: E402

ROOT = os.path.join(HERE, "refdata")


def main():
    m, _ = om.build_model("args/dog_slopes_mixed_args.txt", ROOT)
    p = np.array(m.terrain_params[0][:40])
    terr = {}
    for t in (1, 2, 3, 5, 6, 11, 13):
        for s in (1, 77):
            terr["t_%d_%d" % (t, s)] = om.terrain_build(t, p, s, 20.0)
    np.savez_compressed(os.path.join(HERE, "terrain_golden.npz"), **terr)

    m0, _ = om.build_model("args/sim_dog_args.txt", ROOT)
    e = om.OracleEnv(m0, terrain_seed=5)
    qs, qds = [], []
    for _ in range(12):
        e.step(20); q, qd = e.pose_vel(); qs.append(q); qds.append(qd)
    np.savez_compressed(os.path.join(HERE, "sim_dog_trace.npz"), q=np.array(qs), qd=np.array(qds), terrain_seed=5)

    mr, _ = om.build_model("args/sim_raptor_args.txt", ROOT)
    e = om.OracleEnv(mr, terrain_seed=5)
    qs, qds = [], []
    for _ in range(12):
        e.step(20); q, qd = e.pose_vel(); qs.append(q); qds.append(qd)
    np.savez_compressed(os.path.join(HERE, "sim_raptor_trace.npz"), q=np.array(qs), qd=np.array(qds), terrain_seed=5)

    desc = om.parse_deploy_prototxt(os.path.join(ROOT, "data/policies/dog/nets/dog_mace3_deploy.prototxt"))
    w = om.xavier_weights(desc, 1234)
    io, isc, oo, osc = om.load_scale_file(os.path.join(ROOT, "data/policies/dog/models/dog_mace3_slopes_mixed_model_scale.txt"))
    pol = (desc, w, io, isc, oo, osc)
    mt, _ = om.build_model("args/opt_args_train_mace.txt", ROOT)
    rows, flags, envs = [], [], []
    for i in range(2):
        env = om.OracleEnv(mt, terrain_seed=300 + i, rng_seed=9, env_id=i, policy=pol)
        for _ in range(150):
            env.update()
        r, f = env.drain_tuples(256)
        rows.append(r[:6]); flags.append(f[:6]); envs.append(np.full(len(r[:6]), i))
    np.savez_compressed(os.path.join(HERE, "mace_tuples.npz"), rows=np.concatenate(rows), flags=np.concatenate(flags), env=np.concatenate(envs))

    env = om.OracleEnv(m, terrain_seed=1, policy=pol)
    x = np.random.RandomState(3).uniform(-1, 1, 283)
    np.savez_compressed(os.path.join(HERE, "nn_golden.npz"), x=x, y=env.nn_eval(x))
    print("golden vectors written")


if __name__ == "__main__":
    main()
