#!/usr/bin/env python3
"""Where a MACETrainer.Train() call spends its time on the GPU (synthetic tuples). Run via gpurun."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepterrainrl_amd import trainer as tr
NETS = os.path.join(REPO, "tests/golden/refdata/data/policies/dog/nets")
S, A = 283, 30
rng = np.random.RandomState(0)
def rows(n):
    r = rng.normal(0, 1, size=(n, 1 + 2 * S + A)).astype(np.float32); r[:, 0] = rng.uniform(0, 1, n); r[:, 1 + S] = rng.randint(0, 3, n)
    f = ((rng.uniform(size=n) < 0.5) * 4 + (rng.uniform(size=n) < 0.2)).astype(np.int64)
    return r, f
for graphs in (False, True):
    t = tr.MACETrainer(os.path.join(NETS, "dog_mace3_train.prototxt"), os.path.join(NETS, "dog_mace3_solver.prototxt"), S, A, mem_size=100000, num_init_samples=1000,
                       freeze_target_iters=500, seed=1, use_graphs=graphs)
    r, f = rows(20000); t.AddTuples(r, f)
    for _ in range(20): t.Train()
    torch.cuda.synchronize()
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for k in range(200):
        r, f = rows(32); t.AddTuples(r, f); t.Train()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pr.disable()
    print("graphs=%s: %.2f ms per AddTuples(32)+Train(), actor iters %d" % (graphs, dt / 200 * 1e3, t.actor_iter))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print("\n".join(s.getvalue().splitlines()[6:30]))
