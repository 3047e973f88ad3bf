"""Multi-GPU sharding of the rollout path: one process per GPU, envs split by contiguous GLOBAL env-id ranges.

The rollout itself needs no collective (envs are independent; each reference scene owns its world, ground, character,
controller and RNG: scenarios/ScenarioTrain.cpp:201-221). The only exchange steps of the path are the two the reference
performs between env threads and the trainer:
  (a) experience tuples env -> trainer   (learning/NeuralNetLearner.cpp:33-46)   -> gather_tuples(): counts all-gather +
      gather of fixed-capacity row blocks to the trainer rank (RCCL when the process group backend is "nccl")
  (b) policy weights trainer -> envs     (learning/NeuralNet.cpp:636-658)        -> broadcast_policy(): one broadcast of the
      flat float32 blob (2.28 MB for dog_mace3) + the four normaliser vectors
Both are latency-bound (<= ~10 MB/s at 1 M env-steps/s), so a single collective per outer frame is used instead of bucketing.
Trajectories are shard-invariant: terrain seeds and exploration streams are keyed by the global env id (-global_env_offset=).
"""
import numpy as np


def shard_range(global_envs, world_size, rank):
    """Contiguous split of [0, global_envs) into world_size ranges (first `rem` ranks get one more)."""
    base, rem = divmod(int(global_envs), int(world_size))
    n = base + (1 if rank < rem else 0)
    off = rank * base + min(rank, rem)
    return off, n


class ShardedRollout:
    """One rank's shard of a global batch + the two exchange steps. `dist` is torch.distributed (already initialised)."""

    def __init__(self, make_batch, global_envs, dist=None, device=None):
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.offset, self.n_local = shard_range(global_envs, self.world, self.rank)
        self.global_envs = int(global_envs)
        self.batch = make_batch(self.n_local, self.offset)
        self.device = device
        self.cap = max(2 * max(shard_range(global_envs, self.world, r)[1] for r in range(self.world)), 64)

    def _t(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(self.device) if self.device is not None else t

    def Update(self, dt=1.0 / 30.0):
        self.batch.Update(dt)

    def gather_tuples(self, dst=0):
        """Drain this rank's tuples and gather everybody's on `dst`. Returns (rows, flags, global_env_ids) on dst, else None.
        Layout of a row: MACE replay row [r | s | a | s'] (learning/MACETrainer.cpp:373-401)."""
        import torch
        rows, flags, ids = self.batch.DrainTuples(self.cap)
        o = np.argsort(ids, kind="stable")   # ring order = completion order; sort by env id so the gathered stream does not depend on the sharding
        rows, flags, ids = rows[o], flags[o], ids[o]
        ids = ids.astype(np.int64) + self.offset
        if self.dist is None or self.world == 1:
            return rows, flags, ids
        W = self.batch.W
        n = len(rows)
        # one fixed-capacity block per rank: [count | flags | ids | rows] packed as float64 would waste bandwidth; use 3 small tensors
        cnt = self._t(np.array([n], np.int64))
        counts = [torch.zeros_like(cnt) for _ in range(self.world)]
        self.dist.all_gather(counts, cnt)
        blk_rows = np.zeros((self.cap, W), np.float32); blk_rows[:n] = rows
        blk_meta = np.zeros((self.cap, 2), np.int64); blk_meta[:n, 0] = flags; blk_meta[:n, 1] = ids
        tr, tm = self._t(blk_rows), self._t(blk_meta)
        if self.rank == dst:
            gr = [torch.zeros_like(tr) for _ in range(self.world)]
            gm = [torch.zeros_like(tm) for _ in range(self.world)]
        else:
            gr = gm = None
        self.dist.gather(tr, gr, dst=dst)
        self.dist.gather(tm, gm, dst=dst)
        if self.rank != dst:
            return None
        out_r, out_f, out_i = [], [], []
        for r in range(self.world):
            c = int(counts[r].item())
            out_r.append(gr[r][:c].cpu().numpy()); m = gm[r][:c].cpu().numpy()
            out_f.append(m[:, 0].astype(np.uint32)); out_i.append(m[:, 1])
        return np.concatenate(out_r), np.concatenate(out_f), np.concatenate(out_i)

    def broadcast_policy(self, weights=None, in_off=None, in_scale=None, out_off=None, out_scale=None, src=0):
        """Trainer rank pushes the policy; every rank installs it (cNeuralNetLearner::SyncNet over RCCL)."""
        b = self.batch
        n = b.PolicyNumParams()
        if self.rank == src:
            packed = [np.ascontiguousarray(weights, np.float32)] + [np.ascontiguousarray(a, np.float64) for a in (in_off, in_scale, out_off, out_scale)]
        else:
            packed = [np.zeros(n, np.float32), np.zeros(b.S), np.zeros(b.S), np.zeros(b.nn_out), np.zeros(b.nn_out)]
        if self.dist is not None and self.world > 1:
            ts = [self._t(a) for a in packed]
            for t in ts:
                self.dist.broadcast(t, src=src)
            packed = [t.cpu().numpy() for t in ts]
        b.SetPolicy(*packed)
        return packed
