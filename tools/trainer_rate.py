#!/usr/bin/env python3
"""Trainer iterations per second, isolated from the rollout (run via gpurun): a replay memory of random MACE rows, N x Train() with the target net frozen
(args/opt_args_train_mace.txt: -trainer_freeze_target_iters= 500), for the native HIP step (hip_trainer.HipMACETrainer) and the PyTorch peer
(trainer.MACETrainer, HIP-graph replay). One Train() = critic step + actor candidate filter (+ an actor step whenever 32 candidates have passed).
   python tools/trainer_rate.py [--iters 400] [--rows 20000]"""
import argparse, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from deepterrainrl_amd import trainer as tr, hip_trainer as ht
NETS = os.path.join(REPO, "tests", "golden", "refdata", "data/policies/dog/nets")
TRAIN, SOLVER = (os.path.join(NETS, "dog_mace3_%s.prototxt" % k) for k in ("train", "solver"))
S, A, NF = 283, 30, 3
ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=400); ap.add_argument("--rows", type=int, default=20000); ap.add_argument("--only", default="", help="hip | torch")
ap.add_argument("--repeats", type=int, default=3); ap.add_argument("--lib", default=None, help="another build of the native library (A/B runs)")
a = ap.parse_args()
rng = np.random.RandomState(1)
rows = rng.normal(0, 1, size=(a.rows, 1 + 2 * S + A)).astype(np.float32); rows[:, 0] = rng.uniform(0, 1, a.rows); rows[:, 1 + S] = rng.randint(0, NF, a.rows)
flags = ((rng.uniform(size=a.rows) < 0.4) * 4 + (rng.uniform(size=a.rows) < 0.2) * 1).astype(np.int64)
for name, cls, kw in (("hip (native step)", ht.HipMACETrainer, {"lib_path": a.lib} if a.lib else {}), ("torch peer (HIP graphs)", tr.MACETrainer, {"dtype": torch.float32})):
    if a.only and not name.startswith(a.only):
        continue
    t = cls(TRAIN, SOLVER, S, A, mem_size=1 << 16, num_init_samples=1000, freeze_target_iters=500, device="cuda", seed=3, **kw)
    for k in range(0, a.rows, 4096):
        t.AddTuples(rows[k:k + 4096], flags[k:k + 4096])
    for _ in range(30):
        t.Train()
    rates = []
    for _ in range(a.repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); i0, a0 = t.GetIter(), t.actor_iter
        for _ in range(a.iters):
            t.Train()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rates.append(a.iters / dt)
    print("%-26s %7.1f Train()/s  median of %s (%.3f ms per iteration; last window: %d critic + %d actor solver steps in %d calls; loss %.5f)" % (
        name, float(np.median(rates)), ["%.0f" % r for r in rates], 1e3 / float(np.median(rates)), t.GetIter() - i0, t.actor_iter - a0, a.iters, t.last_loss), flush=True)
