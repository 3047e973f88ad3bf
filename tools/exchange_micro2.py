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
replay = torch.zeros((1 << 18, b.W), device=dev); cur = 0
mode = sys.argv[1]
T = {}
def lap(name, t0):
    if mode == "sync": torch.cuda.synchronize()
    t = time.perf_counter(); T[name] = T.get(name, 0.0) + (t - t0); return t
sr.UpdateBegin()
for k in range(70):
    if k == 20: T.clear(); torch.cuda.synchronize(); tt = time.perf_counter()
    t = time.perf_counter()
    sr.UpdateEnd(); t = lap("UpdateEnd", t)
    if sr._pending is not None:
        g = sr.gather_tuples_end(); t = lap("gend", t)
        m = int(g[0].shape[0])
        if m:
            idx = (torch.arange(m, device=dev) + cur) % replay.shape[0]; t = lap("idx", t)
            replay[idx] = g[0]; cur += m; t = lap("index_put", t)
    # gather_tuples_begin, inlined
    torch.cuda.current_stream(dev).synchronize(); t = lap("stream sync", t)
    n = b.DrainTuplesDevice(sr.stage_rows.data_ptr(), sr.stage_flags.data_ptr(), sr.stage_ids.data_ptr(), sr.cap); t = lap("drain", t)
    W = b.W; blk = sr.block
    hdr = blk[0].view(torch.int32); hdr.zero_(); hdr[0] = n; t = lap("hdr", t)
    if n > 0:
        ids = sr.stage_ids[:n]
        order = torch.argsort(ids, stable=True); t = lap("argsort", t)
        blk[1:n + 1, :W] = sr.stage_rows[:n][order]; t = lap("rows", t)
        meta = blk[1:n + 1, W:].view(torch.int32)
        meta[:, 0] = sr.stage_flags[:n][order]; meta[:, 1] = ids[order] + int(sr.offset); t = lap("meta", t)
    sr._pending = (None, n)
    sr.UpdateBegin(); t = lap("UpdateBegin", t)
torch.cuda.synchronize()
print(mode, "ms per frame %.3f" % ((time.perf_counter() - tt) / 50 * 1e3), {k: round(v / 50 * 1e3, 3) for k, v in T.items()})
