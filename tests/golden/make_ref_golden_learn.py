#!/usr/bin/env python3
"""(CPU, needs oracle/_ref/libref_learn.so = /root/reference/learning compiled) Freezes a run of the REFERENCE'S OWN cMACETrainer for the GPU box, where the HIP trainer
is stepped against it (tests/test_hip_trainer.py::test_gpu_native_trainer_vs_frozen_reference_trainer): the compiled reference trainer (oracle/reflearn.py: its nets are the
numpy fp64 nets of oracle/trainer_ref.py behind cNeuralNet) consumes three batches of seeded tuples and trains 13 iterations with a frozen target (refresh every 2); the
file keeps what a product trainer must reproduce -- every index draw of cMathUtil::gRand, and after every Train(): iteration counters, stage, the three index buffers --
plus a 4000-entry sample and the norm of the final weights. Inputs are regenerated from seeds by the test (rows: numpy RandomState(9); initial weights:
oracle.model.xavier_weights(seed 4321)), so the fixture stays small.

  python tests/golden/make_ref_golden_learn.py      -> tests/golden/ref_golden_learn.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import model as om      # noqa: E402
from oracle import reflearn as rl   # noqa: E402
import test_trainer as TT           # noqa: E402
import test_reference_learn as TR   # noqa: E402

SCHEDULE = ((60, 2), (150, 5), (210, 6))
SEED, W_SEED, FREEZE = 78, 4321, 2


class Recorder:
    def __init__(self, inner):
        self.inner, self.draws = inner, []

    def randint(self, lo, hi=None):
        v = self.inner.randint(lo, hi); self.draws.append(v); return v


def main():
    rng = np.random.RandomState(9)
    rows, flags = TT.random_rows(rng, 420, p_actor=0.5)
    d = om.parse_deploy_prototxt(TT.DEPLOY)
    w0 = om.xavier_weights(d, W_SEED).astype(np.float64)
    H = TR.mace_harness(rl, om)
    R = rl.RefTrainer("mace", H, TT.DEPLOY, TT.SOLVER, mem_size=256, num_init_samples=100, discount=0.9, freeze_target_iters=FREEZE, num_frags=TT.NF, frag_size=TT.FS, seed=SEED)
    for i in range(R.num_pool()):
        R.pool_net(i).w = w0.copy()
    # the draws: a product trainer (torch, CPU, fp64) runs beside the reference on an independent cRand of the same seed -- the lock-step assertions of
    # tests/test_reference_learn.py hold along the way, so its recorded draws ARE the reference's
    t = TT.make_trainer(mem_size=256, num_init_samples=100, seed=21, freeze_target_iters=FREEZE)
    rec = Recorder(rl.RefRandStream(SEED)); t.rng = rec
    t.SetWeights(w0.astype(np.float32))
    steps = []
    k = 0
    for n_new, n_train in SCHEDULE:
        assert R.add_rows(rows[k:k + n_new], flags[k:k + n_new]) == list(t.AddTuples(rows[k:k + n_new], flags[k:k + n_new]))
        k += n_new
        for j in range(n_train):
            R.train(); t.Train()
            assert (R.iter, R.actor_iter, R.stage_train) == (t.GetIter(), t.actor_iter, t.stage_train)
            assert (R.buffer(0), R.buffer(1), R.buffer(2)) == (list(t.critic_buffer), list(t.actor_buffer), list(t.actor_batch_buffer))
            steps.append(dict(iter=R.iter, actor_iter=R.actor_iter, stage=int(R.stage_train), critic=R.buffer(0), actor=R.buffer(1), actor_batch=R.buffer(2)))
    w = R.pool_net(0).w
    assert np.abs(t.net.flat.detach().numpy() - w).max() < 1e-10 * np.abs(w).max()
    pick = np.random.RandomState(1).choice(w.size, 4000, replace=False)
    io, isc = R.input_offset_scale()
    out = dict(draws=np.asarray(rec.draws, np.int32), pick=pick.astype(np.int64), w_pick=w[pick], w_norm=np.linalg.norm(w), w_target_pick=R.pool_net(1).w[pick],
               in_off=io, in_scale=isc, n_steps=len(steps), seed=SEED, w_seed=W_SEED, freeze=FREEZE, schedule=np.asarray(SCHEDULE, np.int32))
    for i, s in enumerate(steps):
        out["s%d_counters" % i] = np.asarray([s["iter"], s["actor_iter"], s["stage"]], np.int32)
        for key in ("critic", "actor", "actor_batch"):
            out["s%d_%s" % (i, key)] = np.asarray(s[key], np.int32)
    path = os.path.join(HERE, "ref_golden_learn.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(rec.draws), "draws, final iter", R.iter, "actor iter", R.actor_iter)


if __name__ == "__main__":
    main()
