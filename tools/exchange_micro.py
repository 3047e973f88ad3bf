#!/usr/bin/env python3
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, REPO)
import torch, numpy as np
import bench
import deepterrainrl_amd as da
from deepterrainrl_amd.sharding import ShardedRollout
dev = torch.device("cuda", 0)
make = lambda nl, off: da.BatchScenario(bench.EXCHANGE_ARG_FILE, nl, data_root=bench.ROOT, extra_args={"terrain_seed": 20260925, "rand_seed": 1, "global_env_offset": off})
sr = ShardedRollout(make, 4096, device=dev); b = sr.batch
sr.broadcast_policy(bench.xavier_weights(b.PolicyNumParams()), *bench.load_scale()); b.SetExplore(True, 0.2, 0.025, 0.002)
for k in range(25): b.Update(); b.DrainTuples()
def tm(name, fn, reps=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); print("%-28s %.3f ms" % (name, (time.perf_counter() - t0) / reps * 1e3)); return r
for it in range(3):
    b.Update()
    n = tm("DrainTuplesDevice", lambda: b.DrainTuplesDevice(sr.stage_rows.data_ptr(), sr.stage_flags.data_ptr(), sr.stage_ids.data_ptr(), sr.cap))
    print("n =", n)
    blk = sr.block; W = b.W
    hdr = blk[0].view(torch.int32)
    tm("hdr.zero_", lambda: hdr.zero_())
    def sethdr(): hdr[0] = n
    tm("hdr[0]=n", sethdr)
    ids = sr.stage_ids[:n]
    order = tm("argsort", lambda: torch.argsort(ids, stable=True))
    def rows(): blk[1:n + 1, :W] = sr.stage_rows[:n][order]
    tm("rows gather+copy", rows)
    meta = blk[1:n + 1, W:].view(torch.int32)
    def m0(): meta[:, 0] = sr.stage_flags[:n][order]
    tm("meta0", m0)
    def m1(): meta[:, 1] = ids[order] + 0
    tm("meta1", m1)
    cnt = tm("tolist", lambda: torch.stack([blk[0].view(torch.int32)[0]]).tolist())
    r = tm("cat", lambda: torch.cat([blk[1:n + 1, :W]]))
    replay = torch.zeros((1 << 18, W), device=dev)
    def app():
        idx = (torch.arange(n, device=dev) + 5) % replay.shape[0]; replay[idx] = r
    tm("append", app)
    tm("empty sync", lambda: None)
    tm("UpdateBegin+End", lambda: (b.UpdateBegin(), b.UpdateEnd()))
