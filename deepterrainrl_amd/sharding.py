"""Multi-GPU sharding of the rollout path: one process per GPU, envs split by contiguous GLOBAL env-id ranges.

The rollout itself needs no collective (envs are independent; each reference scene owns its world, ground, character,
controller and RNG: scenarios/ScenarioTrain.cpp:201-221). The only exchange steps of the path are the two the reference
performs between env threads and the trainer:
  (a) experience tuples env -> trainer   (learning/NeuralNetLearner.cpp:33-46)
  (b) policy weights trainer -> envs     (learning/NeuralNet.cpp:636-658)
Both are latency-bound on xGMI (<= ~10 MB/s of tuples at 1 M env-steps/s; 2.3 MB of weights per push), so each is ONE collective on ONE
preallocated device buffer, issued on a side stream so that it overlaps the next frame kernel:

  (a) gather_tuples_begin(): dtrl_drain_tuples_packed moves the frame's rows inside the engine (order / copy / header kernels) into this
      rank's block [block_rows + 1, W + 2] float32 (row 0 = header carrying the count; the two extra columns carry the flag word and the
      GLOBAL env id as raw int32 bits), sorted by env id (so the gathered stream does not depend on how envs are sharded), and starts one
      GATHER of the block to the trainer rank (RCCL: ncclSend / ncclRecv group over xGMI) on the comm stream -- only the trainer consumes tuples, so
      nobody else receives any. The block is sized for the steady state, not for the worst case: block_rows = max(64, envs per rank / 8) rows
      (a frame completes ~0.08 tuples per env: one per gait cycle of ~12.5 frames; 4096 envs: 513 x 599 floats = 1.2 MB per rank and frame, where
      round 2 all-gathered a 2 N-row block of 19.6 MB to every rank). Rows that do not fit stay in the engine's ring, in order, and travel with a
      later frame (the batch starts in lock-step, so the first tuple frames are bursts); the header reports how many were carried.
      gather_tuples_end() waits and hands the trainer rank the concatenated DEVICE tensors (rows, flags, ids) -- nothing visits the host.
      gather_tuples() = begin + end.
  (b) broadcast_policy(): weights (float32) and the four normaliser vectors (float64) travel as ONE byte buffer (one ncclBroadcast), and
      every rank installs them with dtrl_set_policy_device (a gather kernel re-lays the blob; no host round trip).

With a CPU process group (gloo, tests/test_multi_gpu_gloo.py, lane-loop test backend) the same code runs on CPU tensors: "device" pointers
are host pointers there. Trajectories are shard-invariant: terrain seeds and exploration streams are keyed by the global env id
(-global_env_offset=).
"""
import numpy as np


def shard_range(global_envs, world_size, rank):
    """Contiguous split of [0, global_envs) into world_size ranges (first `rem` ranks get one more)."""
    base, rem = divmod(int(global_envs), int(world_size))
    n = base + (1 if rank < rem else 0)
    off = rank * base + min(rank, rem)
    return off, n


def default_block_rows(global_envs, world_size):
    """Rows of a rank's send block: the steady state produces ~0.08 tuples per env and frame; an eighth of the largest shard (at least 64 rows) leaves
    ~50 % head room, and what does not fit is carried to a later frame by the engine (never dropped)."""
    n_max = -(-int(global_envs) // int(world_size))
    return max(64, -(-n_max // 8))


_STAGED_LOGGED = False
_PIN = {}


def host_staged(dist, t):
    """gloo's collectives are taken with host tensors here (two ranks sharing one GPU on a one-GPU box, debugging without RCCL): a device tensor goes through
    the host around the collective. Never true with RCCL (backend "nccl") or with the CPU test backends. The first time it IS true a line goes to stderr: it is a
    debug path (a host sync per collective), and check_backend_for_devices() refuses it outright when the ranks sit on different GPUs."""
    global _STAGED_LOGGED
    on = dist is not None and bool(getattr(t, "is_cuda", False)) and dist.get_backend() == "gloo"
    if on and not _STAGED_LOGGED:
        import sys
        _STAGED_LOGGED = True
        print("[dtrl] collectives over DEVICE tensors on a gloo group: staged through pinned host memory (one-GPU debug path, a host sync per collective; "
              "use the nccl backend = RCCL on a multi-GPU node)", file=sys.stderr)
    return on


def _pinned_like(t):
    """one page-locked staging buffer per (shape, dtype), reused (the gradient all-reduce of the data-parallel trainer runs twice per Train())"""
    import torch
    key = (tuple(t.shape), t.dtype)
    h = _PIN.get(key)
    if h is None:
        h = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        _PIN[key] = h
    return h


def check_backend_for_devices(dist, device):
    """Start-up guard (VERDICT r5 #6b): with more than one rank and exchange buffers on GPUs, the backend must be nccl (= RCCL) unless the ranks SHARE one device
    (the one-GPU box's two-rank tests): the host-staged gloo path can never be picked silently on a real node. Returns True when host staging will be used."""
    import os
    import torch
    if dist is None or dist.get_world_size() < 2 or device is None or torch.device(device).type != "cuda":
        return False
    if dist.get_backend() == "nccl":
        return False
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    try:
        uid = str(torch.cuda.get_device_properties(idx).uuid)
    except Exception:
        uid = "%s:%d" % (os.uname().nodename, idx)
    mine = (os.uname().nodename, os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", "")), idx, uid)
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    if len(set(everyone)) > 1:
        raise RuntimeError("ShardedRollout: %d ranks on DIFFERENT GPUs %s with backend '%s': the exchange must run over RCCL (init_process_group(backend='nccl')); the "
                           "host-staged gloo path is a one-GPU debug path" % (dist.get_world_size(), sorted(set(everyone)), dist.get_backend()))
    return True


def all_reduce(dist, t, op=None):
    """dist.all_reduce(t) on the CURRENT stream; staged through the host for a gloo group over device tensors (the blocking copies order it on that stream)."""
    kw = {} if op is None else {"op": op}
    if host_staged(dist, t):
        h = _pinned_like(t); h.copy_(t); dist.all_reduce(h, **kw); t.copy_(h)
    else:
        dist.all_reduce(t, **kw)


def broadcast(dist, t, src):
    if host_staged(dist, t):
        h = _pinned_like(t); h.copy_(t); dist.broadcast(h, src=src); t.copy_(h)
    else:
        dist.broadcast(t, src=src)


class ShardedRollout:
    """One rank's shard of a global batch + the two exchange steps. `dist` is torch.distributed (already initialised) or None.
    `device`: torch device of the exchange buffers -- the GPU the batch runs on (RCCL; with a gloo group the collectives are staged through
    pinned host memory), or None / "cpu" with a gloo group and the lane-loop test backend.
    `block_rows`: rows per rank and gather (default_block_rows); every rank must pass the same value."""

    def __init__(self, make_batch, global_envs, dist=None, device=None, force_collectives=None, pipelined=False, block_rows=None):
        import os
        import torch
        self.torch = torch
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        # collectives run when there is more than one rank; force_collectives (or DTRL_FORCE_COLLECTIVES=1) runs them on a one-rank group as
        # well, which is how the RCCL code path is exercised on a single-GPU box
        if force_collectives is None:
            force_collectives = os.environ.get("DTRL_FORCE_COLLECTIVES") == "1"
        self.coll = dist is not None and (self.world > 1 or bool(force_collectives))
        self.offset, self.n_local = shard_range(global_envs, self.world, self.rank)
        self.global_envs = int(global_envs)
        self.batch = make_batch(self.n_local, self.offset)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.on_gpu = self.device.type == "cuda"
        b = self.batch
        # one block size for every rank (a fixed-size collective), derived from the global shape alone: no start-up exchange needed
        self.cap = int(block_rows) if block_rows else default_block_rows(global_envs, self.world)
        W = b.W
        # pipelined: the engine switches between two tuple rings at every UpdateBegin, so frame f's tuples are drained and gathered WHILE frame f + 1 runs:
        # UpdateEnd (f) -> gather_tuples_end (f - 1) -> UpdateBegin (f + 1) -> gather_tuples_begin (f). Same tuple stream as the sequential protocol.
        self.pipelined = bool(pipelined)
        if self.pipelined:
            b.SetTuplePipelining(True)
        # this rank's send block: header row + cap tuple rows, W floats + [flags, global env id] as int32 bit patterns. Pipelined on a GPU: two of them, used
        # alternately, so that the drain of frame f never has to wait (on the host) for the readers of frame f - 1's block. A block is rewritten only after
        # the event recorded BEHIND its last reader has fired (self._busy): the gather that sent it (recorded once this rank's stream has waited for the
        # collective -- an event on the comm stream right after an async collective call fires before RCCL's kernel has read the block), or the consumer of
        # the views a one-rank run hands out.
        nblk = 2 if (self.pipelined and self.on_gpu) else 1
        self.blocks = [torch.zeros((self.cap + 1, W + 2), dtype=torch.float32, device=self.device) for _ in range(nblk)]
        self.block = self.blocks[0]
        self._bi = 0
        self._busy = [None] * nblk
        self._handed = None        # index of the block whose views the last gather_tuples_end handed to the caller (one-rank runs)
        self.gathered = None       # receive side, trainer rank only: allocated by the first gather_tuples_end(dst) on that rank
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        # a gloo group over DEVICE buffers (two ranks sharing one GPU on a box without a second one; debugging without RCCL): gloo's gather takes host
        # tensors only, so the block / the policy buffer are staged through pinned host memory around the collective. Never taken with RCCL.
        self.staged = bool(self.coll and self.on_gpu and dist.get_backend() == "gloo")
        if self.coll and self.on_gpu:
            check_backend_for_devices(dist, self.device)     # raises when ranks on different GPUs would go through the host
        self._stage = [torch.zeros((self.cap + 1, W + 2), dtype=torch.float32).pin_memory() for _ in range(nblk)] if self.staged else None
        self._stage_recv = None
        self._pending = None
        self.carried_rows = 0      # rows the engine kept back for a later frame because the block was full (sum of the headers seen on this rank)
        # policy buffer: [weights f32 | in_off | in_scale | out_off | out_scale f64], 8-byte aligned sections
        self.n_w = b.PolicyNumParams()
        self.w_bytes = (4 * self.n_w + 7) // 8 * 8
        self.pol_bytes = self.w_bytes + 8 * (2 * b.S + 2 * b.nn_out)
        self.pol_buf = torch.zeros(self.pol_bytes, dtype=torch.uint8, device=self.device) if self.n_w else None
        self.exchange_wait_s = 0.0

    @property
    def block_bytes(self):
        return int(self.block.numel() * 4)

    # ---- rollout ----
    def Update(self, dt=1.0 / 30.0):
        self.batch.Update(dt)

    def UpdateBegin(self, dt=1.0 / 30.0):
        self.batch.UpdateBegin(dt)

    def UpdateEnd(self):
        self.batch.UpdateEnd()

    def UpdateEndBegin(self, dt=1.0 / 30.0):
        self.batch.UpdateEndBegin(dt)

    # ---- (a) experience tuples ----
    def gather_tuples_begin(self, dst=0):
        """Drain this rank's finished tuples into its block and start the gather to rank `dst` (asynchronous on the GPU). Sequential protocol: call between
        UpdateEnd() of frame f and UpdateBegin() of frame f + 1; the collective then overlaps frame f + 1's kernel. Pipelined protocol (pipelined=True):
        call right AFTER UpdateBegin() of frame f + 1 -- the drain itself (order / copy kernels on the engine's drain stream) overlaps that frame too. Collect it with gather_tuples_end() in the
        NEXT gap (after UpdateEnd() of frame f + 1): a frame kernel fills every CU, so small kernels and host syncs issued while it runs stall
        until it ends (bench.py's exchange leg: UpdateEndBegin -> gather_tuples_end (previous frame) -> consume -> gather_tuples_begin).
        The packing (sort by env id so that the gathered stream does not depend on how envs are sharded, flag word and GLOBAL env id appended to every
        row, header row with the count) happens inside the engine (dtrl_drain_tuples_packed): no framework op touches the rows."""
        torch = self.torch
        assert self._pending is None, "gather_tuples_begin called twice without gather_tuples_end"
        if self.on_gpu and self._handed is not None:
            # one-rank run: the caller reads views of that block on ITS stream; everything it has queued so far comes before the block's next rewrite
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(self.device)); self._busy[self._handed] = ev
            self._handed = None
        if len(self.blocks) > 1:
            self._bi ^= 1
        bi = self._bi
        self.block = self.blocks[bi]
        if self._busy[bi] is not None:
            self._busy[bi].synchronize()                                     # pipelined: two frames old, fired long ago
            self._busy[bi] = None
        self.batch.DrainTuplesPacked(self.block.data_ptr(), self.cap)        # synchronised: the block is complete when this returns
        work = None
        if self.coll:
            recv = None
            if self.rank == dst:
                if self.gathered is None:
                    self.gathered = [torch.zeros_like(self.block) for _ in range(self.world)]
                recv = self.gathered
            if self.staged:
                self._stage[bi].copy_(self.block)                                       # blocking copy: the device block is free again when it returns
                if self.rank == dst and self._stage_recv is None:
                    self._stage_recv = [torch.zeros_like(self._stage[0]) for _ in range(self.world)]
                work = self.dist.gather(self._stage[bi], gather_list=self._stage_recv if self.rank == dst else None, dst=dst, async_op=True)
            elif self.on_gpu:
                self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))   # readers of the previous gather result (queued on the caller's stream) come first
                with torch.cuda.stream(self.comm_stream):
                    work = self.dist.gather(self.block, gather_list=recv, dst=dst, async_op=True)
            else:
                work = self.dist.gather(self.block, gather_list=recv, dst=dst, async_op=True)
        self._pending = (work, bi, dst)

    def gather_tuples_end(self, dst=None, want_meta=True):
        """Wait for the gather. Returns (rows [n, W] float32, flags [n] int32, global env ids [n] int32) as tensors on self.device on the rank the
        gather was started towards, None elsewhere; want_meta=False returns (rows, None, None) on a multi-rank group."""
        import time
        torch = self.torch
        work, bi, dst0 = self._pending
        assert dst is None or dst == dst0, "gather_tuples_end(dst) differs from gather_tuples_begin(dst)"
        dst = dst0
        self._pending = None
        W = self.batch.W
        t0 = time.perf_counter()
        if work is not None:
            work.wait()
            if self.staged:
                if self.rank == dst:
                    for g, h in zip(self.gathered, self._stage_recv):
                        g.copy_(h)
            elif self.on_gpu:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_stream(self.comm_stream)
                # behind the collective itself: this rank's stream now follows RCCL's kernel, so an event recorded here fires only after the send block was read
                ev = torch.cuda.Event(); ev.record(cur); self._busy[bi] = ev
        self.exchange_wait_s += time.perf_counter() - t0
        if self.rank != dst:
            return None
        if not self.coll:
            blk_t = self.blocks[bi]
            hdr = blk_t[0, :3].view(torch.int32).tolist()             # one 12-byte read-back; the rows below are views into the block (valid until its next begin)
            c = int(hdr[0]); self.carried_rows += int(hdr[2])
            blk = blk_t[1:c + 1]
            meta = blk[:, W:].view(torch.int32)
            self._handed = bi
            return blk[:, :W], meta[:, 0], meta[:, 1]
        blocks = self.gathered
        hdr = torch.stack([g[0, :3].view(torch.int32) for g in blocks]).tolist()
        counts = [int(h[0]) for h in hdr]
        self.carried_rows += sum(int(h[2]) for h in hdr)
        rows = torch.cat([g[1:c + 1, :W] for g, c in zip(blocks, counts)])
        if not want_meta:                 # (every framework op issued while a frame kernel runs waits for a wavefront slot: a consumer of the rows alone skips three)
            return rows, None, None
        meta = torch.cat([g[1:c + 1, W:].view(torch.int32) for g, c in zip(blocks, counts)])
        return rows, meta[:, 0].contiguous(), meta[:, 1].contiguous()

    def gather_tuples(self, dst=0):
        """Synchronous form. Returns numpy (rows, flags uint32, global env ids int64) on dst, else None (the host-side trainer loop's interface)."""
        self.gather_tuples_begin(dst)
        g = self.gather_tuples_end(dst)
        if g is None:
            return None
        rows, flags, ids = g
        return rows.cpu().numpy(), flags.cpu().numpy().astype(np.uint32), ids.cpu().numpy().astype(np.int64)

    # ---- (b) policy ----
    def _pol_views(self):
        torch = self.torch
        b = self.batch
        buf = self.pol_buf
        w = buf[:4 * self.n_w].view(torch.float32)
        o = self.w_bytes
        secs = []
        for n in (b.S, b.S, b.nn_out, b.nn_out):
            secs.append(buf[o:o + 8 * n].view(torch.float64)); o += 8 * n
        return [w] + secs

    def broadcast_policy(self, weights=None, in_off=None, in_scale=None, out_off=None, out_scale=None, src=0, normalizers=True, want_host=True):
        """Trainer rank pushes the policy (numpy arrays or tensors); every rank installs it from the device buffer
        (cNeuralNetLearner::SyncNet over one RCCL broadcast). Returns the five arrays as numpy (host copies for callers that log them; want_host=False: None).
        normalizers=False (every rank must pass the same value) installs the weights only -- which, while a frame is in flight, does not wait for it
        (dtrl_set_policy_device: second weight buffer, switched in by the next launch); the four vectors still travel in the one buffer, unused."""
        torch = self.torch
        views = self._pol_views()
        if self.rank == src:
            for v, a in zip(views, (weights, in_off, in_scale, out_off, out_scale)):
                if a is None:
                    continue
                t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, np.float32 if v.dtype == torch.float32 else np.float64))
                v.copy_(t.to(self.device, dtype=v.dtype).reshape(-1))
        if self.coll and self.staged:
            host = self.pol_buf.cpu()
            self.dist.broadcast(host, src=src)
            self.pol_buf.copy_(host)
        elif self.coll:
            self.dist.broadcast(self.pol_buf, src=src)
        if not normalizers:
            # no host wait: the re-layout kernel is queued on this stream behind the broadcast (a synchronous-mode collective blocks the CURRENT stream on RCCL's),
            # the engine's next frame launches wait for it on the device (dtrl_set_policy_device_async). pol_buf's next writer is the next broadcast_policy, queued
            # on the same stream. (Round 3 synchronised the stream here: a host wait behind the frame kernel in flight, once per hand-over.)
            st = torch.cuda.current_stream(self.device).cuda_stream if self.on_gpu else None
            self.batch.SetPolicyDeviceAsync(views[0].data_ptr(), self.n_w, st)
            return [v.cpu().numpy().copy() for v in views] if want_host else None
        if self.on_gpu:
            torch.cuda.current_stream(self.device).synchronize()
        self.batch.SetPolicyDevice(views[0].data_ptr(), self.n_w, *[v.data_ptr() for v in views[1:]])
        return [v.cpu().numpy().copy() for v in views] if want_host else None
