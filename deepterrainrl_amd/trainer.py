"""MACE trainer on the GPU: the learner side of the rollout path, mirroring cMACETrainer / cNeuralNetTrainer.

What the reference does per call and where it lives here:
  cNeuralNetTrainer::AddTuple  (learning/NeuralNetTrainer.cpp:145-165, CheckTuple :541-576)   -> MACETrainer.AddTuples
  cMACETrainer::UpdateBuffers  (learning/MACETrainer.cpp:731-799)                              -> MACETrainer._update_buffers
  cNeuralNetTrainer::Train / UpdateStage / InitStage / UpdateOffsetScale (:175-183, 696-747)   -> MACETrainer.Train
  cMACETrainer::Step           (learning/MACETrainer.cpp:346-372)                              -> MACETrainer.Step
  critic targets BuildProblemY / CalcNewCumulativeRewardBatch (:226-250, 478-515)              -> MACETrainer._critic_problem
  actor UpdateActorBatchBuffer / UpdateActor / StepActor (:577-633), BuildActorProblemY (:285-305)
  cNeuralNet::LoadTrainData / Eval normalisation (learning/NeuralNet.cpp:1077-1122, 352-375, 964-1036)
  Caffe SGDSolver step (base_lr, momentum, weight_decay from the solver prototxt; lr_mult / decay_mult per blob from the train
  prototxt; EuclideanLoss = 1/(2N) sum ||y - label||^2): L2 regularise, history = momentum * history + lr * diff, w -= history.

The replay memory, both networks (current + frozen target), the solver history and every batch stay resident on the device
(torch tensors: PyTorch provides memory, autograd and the conv/GEMM calls); the host only keeps the index buffers. Rows arrive in
the exact MACE replay layout the rollout engine emits ([r | s | a | s'], float32), so `dtrl_drain_tuples` output is appended as is.
The only randomness is minibatch sampling (the reference uses its global RNG; a seeded numpy stream here).
"""
import os
import re

import numpy as np
import torch

FLAG_FAIL, FLAG_EXP_CRITIC, FLAG_EXP_ACTOR = 1, 2, 4   # tExpTuple flag bits as emitted by the rollout engine (include/dtrl.h)
FLAG_OFF_POLICY = 2                                    # cCaclaTrainer::eFlagOffPolicy (the CACLA scenes use bit 1 for it)


def _blocks(txt):
    """Split a Caffe prototxt (or the fixture digest of one) into layer blocks with balanced braces."""
    out = []
    for m in re.finditer(r"\blayer\s*\{", txt):
        depth, i = 1, m.end()
        while depth and i < len(txt):
            depth += {"{": 1, "}": -1}.get(txt[i], 0)
            i += 1
        out.append(txt[m.end():i - 1])
    return out


def parse_net(path):
    """Topology + per-blob lr_mult / decay_mult of the MACE family (slice -> 3 conv1d -> terr_ip0 -> concat -> ip0 -> val / a* heads)."""
    txt = open(path).read()
    dims = [int(x) for x in re.findall(r"input_dim:\s*(\d+)", txt)]
    layers = []
    for blk in _blocks(txt):
        g = lambda key, cast=int: (lambda m: cast(m.group(1)) if m else None)(re.search(key + r":\s*([-\d.eE]+)", blk))
        mults = [(float((re.search(r"lr_mult:\s*([-\d.eE]+)", p) or [0, 1])[1]), float((re.search(r"decay_mult:\s*([-\d.eE]+)", p) or [0, 1])[1]))
                 for p in re.findall(r"\bparam\s*\{([^}]*)\}", blk)]
        layers.append({"name": re.search(r'name:\s*"([^"]+)"', blk).group(1), "type": re.search(r'type:\s*"([^"]+)"', blk).group(1),
                       "num_output": g("num_output"), "kernel_w": g("kernel_w"), "slice_point": g("slice_point"),
                       "batch_size": g("batch_size"), "width": g("width"), "mults": mults})
    d = {"layers": layers}
    data = [l for l in layers if l["type"] == "MemoryData"]
    d["in_size"] = dims[-1] if dims else data[0]["width"]
    d["batch_size"] = data[0]["batch_size"] if data else 1
    d["n_terrain"] = [l for l in layers if l["type"] == "Slice"][0]["slice_point"]
    d["convs"] = [l for l in layers if l["type"] == "Convolution"]
    d["ips"] = {l["name"]: l for l in layers if l["type"] == "InnerProduct"}
    if "val_ip1" in d["ips"]:                                  # MACE family: value head + n_frags actor heads
        d["n_frags"] = d["ips"]["val_ip1"]["num_output"]
        d["frag_size"] = d["ips"]["a0_ip1"]["num_output"]
    else:                                                      # single-head family (Q net, CACLA actor): terr_ip0 -> ip1 -> ip2 -> output
        d["n_frags"] = 0
        d["frag_size"] = d["ips"]["output"]["num_output"]
    return d


def parse_solver(path):
    txt = open(path).read()
    g = lambda key, dflt: (lambda m: float(m.group(1)) if m else dflt)(re.search(r"^\s*" + key + r":\s*([-\d.eE]+)", txt, re.M))
    pol = re.search(r'lr_policy:\s*"([^"]+)"', txt)
    return {"base_lr": g("base_lr", 0.01), "momentum": g("momentum", 0.0), "weight_decay": g("weight_decay", 0.0),
            "lr_policy": pol.group(1) if pol else "fixed", "gamma": g("gamma", 0.1), "stepsize": g("stepsize", 1e18), "power": g("power", 0.75)}


class MaceNet(torch.nn.Module):
    """The dog/raptor *_mace3 net. Blob order (= the flat weight vector the rollout engine takes): conv0..2, terr_ip0, ip0,
    val_ip0, val_ip1, a{f}_ip0, a{f}_ip1; weight then bias. Works in normalised input/output space like the Caffe net."""

    def __init__(self, desc):
        super().__init__()
        self.n_terrain = desc["n_terrain"]
        n_char = desc["in_size"] - self.n_terrain
        mods, mults = [], []
        cin, w = 1, self.n_terrain
        for l in desc["convs"]:
            mods.append(torch.nn.Conv1d(cin, l["num_output"], l["kernel_w"])); mults.append(l["mults"])
            cin, w = l["num_output"], w - l["kernel_w"] + 1
        ips = desc["ips"]
        def lin(name, nin):
            mods.append(torch.nn.Linear(nin, ips[name]["num_output"])); mults.append(ips[name]["mults"])
            return ips[name]["num_output"]
        n = lin("terr_ip0", cin * w)
        n = lin("ip0", n + n_char)
        nh = lin("val_ip0", n); lin("val_ip1", nh)
        self.n_frags = desc["n_frags"]
        for f in range(self.n_frags):
            nh = lin("a%d_ip0" % f, n); lin("a%d_ip1" % f, nh)
        self.mods = torch.nn.ModuleList(mods)
        # (lr_mult, decay_mult) per blob, Caffe defaults 1 / 1 when a param block or a field is absent
        self.blob_mults = []
        for m in mults:
            m = list(m) + [(1.0, 1.0)] * (2 - len(m))
            self.blob_mults += [m[0], m[1]]
        self.flat = self.gflat = None

    def flatten_storage(self):
        """Re-home every blob (and its gradient) as a view into ONE flat buffer in the engine's blob order: the solver update, the
        target-net copy and the weight hand-over to the rollout engine become single tensor ops instead of 26 small ones."""
        blobs = self.blobs()
        n = sum(b.numel() for b in blobs)
        self.flat = torch.empty(n, dtype=blobs[0].dtype, device=blobs[0].device)
        self.gflat = torch.zeros_like(self.flat)
        off = 0
        for b in blobs:
            k = b.numel()
            self.flat[off:off + k].copy_(b.detach().reshape(-1))
            b.data = self.flat[off:off + k].view(b.shape)
            b.grad = self.gflat[off:off + k].view(b.shape)
            off += k
        return self

    def init_fillers(self, seed):
        """Caffe fillers of the train prototxt: weights "xavier" (uniform +-sqrt(3 / fan_in)), biases "constant" 0; seeded, so two
        trainers built with the same seed start from the same policy (the reference draws from Caffe's RNG)."""
        g = torch.Generator().manual_seed(int(seed))
        with torch.no_grad():
            for m in self.mods:
                fan_in = m.weight[0].numel()
                s = (3.0 / fan_in) ** 0.5
                m.weight.copy_(((torch.rand(m.weight.shape, generator=g, dtype=torch.float64) * 2 - 1) * s).to(m.weight.dtype))
                m.bias.zero_()
        return self

    def blobs(self):
        out = []
        for m in self.mods:
            out += [m.weight, m.bias]
        return out

    def forward(self, x):
        relu = torch.nn.functional.relu
        t = x[:, :self.n_terrain].unsqueeze(1)
        for m in self.mods[:3]:
            t = relu(m(t))
        t = relu(self.mods[3](t.flatten(1)))
        h = relu(self.mods[4](torch.cat([t, x[:, self.n_terrain:]], 1)))
        outs = [self.mods[6](relu(self.mods[5](h)))]
        for f in range(self.n_frags):
            outs.append(self.mods[8 + 2 * f](relu(self.mods[7 + 2 * f](h))))
        return torch.cat(outs, 1)

    def named_blobs(self, x):
        """Forward pass that keeps every Caffe blob of the deploy net by name (cNeuralNet::GetLayerState reads mNet->blob_by_name): the ReLU layers
        of these nets write to their own tops, so "terr_conv0" is the pre-activation and "terr_relu0" the rectified blob. x: normalised input [B, S]."""
        relu = torch.nn.functional.relu
        out = {"data": x, "data_terrain": x[:, :self.n_terrain], "data_char": x[:, self.n_terrain:]}
        t = x[:, :self.n_terrain].unsqueeze(1)
        for l in range(3):
            t = self.mods[l](t); out["terr_conv%d" % l] = t.flatten(1)
            t = relu(t); out["terr_relu%d" % l] = t.flatten(1)
        t = self.mods[3](t.flatten(1)); out["terr_ip0"] = t
        t = relu(t); out["terr_relu3"] = t
        out["char_flatten0"] = x[:, self.n_terrain:]
        c = torch.cat([t, x[:, self.n_terrain:]], 1); out["concat0"] = c
        h = self.mods[4](c); out["ip0"] = h
        h = relu(h); out["relu0"] = h
        heads = []
        for i, pre in enumerate(["val"] + ["a%d" % f for f in range(self.n_frags)]):
            z = self.mods[5 + 2 * i](h); out[pre + "_ip0"] = z
            z = relu(z); out[pre + "_relu0"] = z
            y = self.mods[6 + 2 * i](z); out[pre + "_ip1"] = y
            heads.append(y)
        out["output"] = torch.cat(heads, 1)
        return out

    def get_flat(self):
        src = self.flat if self.flat is not None else torch.cat([b.detach().reshape(-1) for b in self.blobs()])
        return src.detach().to(torch.float32).cpu().numpy()

    def set_flat(self, w):
        w = torch.as_tensor(np.asarray(w), dtype=self.mods[0].weight.dtype, device=self.mods[0].weight.device)
        off = 0
        with torch.no_grad():
            for b in self.blobs():
                b.copy_(w[off:off + b.numel()].reshape(b.shape)); off += b.numel()
        assert off == w.numel(), "weight count does not match the net"

    def num_params(self):
        return sum(b.numel() for b in self.blobs())


class QNet(MaceNet):
    """The single-head nets (data/policies/dog/nets/dog_q_*.prototxt; same layers as the CACLA actor): 3 conv1d over the terrain slice -> terr_ip0 ->
    concat with the character slice -> ip1 -> ip2 -> output (one value per base action). Blob order conv0..2, terr_ip0, ip1, ip2, output."""

    def __init__(self, desc):
        torch.nn.Module.__init__(self)
        self.n_terrain = desc["n_terrain"]
        n_char = desc["in_size"] - self.n_terrain
        mods, mults = [], []
        cin, w = 1, self.n_terrain
        for l in desc["convs"]:
            mods.append(torch.nn.Conv1d(cin, l["num_output"], l["kernel_w"])); mults.append(l["mults"])
            cin, w = l["num_output"], w - l["kernel_w"] + 1
        ips = desc["ips"]
        n = cin * w
        for name, extra in (("terr_ip0", 0), ("ip1", n_char), ("ip2", 0), ("output", 0)):
            mods.append(torch.nn.Linear(n + extra, ips[name]["num_output"])); mults.append(ips[name]["mults"])
            n = ips[name]["num_output"]
        self.n_frags = 0
        self.mods = torch.nn.ModuleList(mods)
        self.blob_mults = []
        for m in mults:
            m = list(m) + [(1.0, 1.0)] * (2 - len(m))
            self.blob_mults += [m[0], m[1]]
        self.flat = self.gflat = None

    def forward(self, x):
        relu = torch.nn.functional.relu
        t = x[:, :self.n_terrain].unsqueeze(1)
        for m in self.mods[:3]:
            t = relu(m(t))
        t = relu(self.mods[3](t.flatten(1)))
        h = relu(self.mods[4](torch.cat([t, x[:, self.n_terrain:]], 1)))
        return self.mods[6](relu(self.mods[5](h)))

    def named_blobs(self, x):
        raise NotImplementedError("blob names of the single-head nets are not mapped")


class MACETrainer:
    """cMACETrainer (pool size 1, synchronous mode)."""

    def __init__(self, net_file, solver_file, state_size, action_size, mem_size=500000, num_init_samples=200, steps_per_iter=1,
                 freeze_target_iters=0, discount=0.9, init_input_offset_scale=True, device=None, dtype=torch.float32, seed=0, use_graphs=None):
        self.desc = parse_net(net_file)
        self.solver = parse_solver(solver_file)
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.dtype = dtype
        self.S, self.A = state_size, action_size
        self.num_frags, self.frag_size = self.desc["n_frags"], self.desc["frag_size"]
        self._configure_sizes()
        self.batch = self.desc["batch_size"]
        self.W = 1 + 2 * state_size + action_size                      # cMACETrainer::CalcBufferSize
        self.mem_size, self.num_init_samples, self.steps_per_iter = mem_size, num_init_samples, steps_per_iter
        self.freeze_target_iters, self.discount, self.init_input_offset_scale = freeze_target_iters, discount, init_input_offset_scale
        self.net = self._new_net().init_fillers(seed).to(self.device, dtype).flatten_storage()
        self.target = self._new_net().to(self.device, dtype).flatten_storage()
        for p_ in self.target.parameters():
            p_.requires_grad_(False)
        self.hflat = torch.zeros_like(self.net.flat)
        self.history, self.rate_mult, self.decay_mult = [], torch.empty_like(self.hflat), torch.empty_like(self.hflat)   # per-element lr_mult / decay_mult
        off = 0
        for b, (lm, dm) in zip(self.net.blobs(), self.net.blob_mults):
            k = b.numel()
            self.history.append(self.hflat[off:off + k].view(b.shape)); self.rate_mult[off:off + k] = lm; self.decay_mult[off:off + k] = dm
            off += k
        z = lambda n, v: torch.full((n,), v, device=self.device, dtype=dtype)
        self.in_off, self.in_scale, self.out_off, self.out_scale = z(state_size, 0.0), z(state_size, 1.0), z(self.out_size, 0.0), z(self.out_size, 1.0)
        self.mem = torch.zeros((mem_size, self.W), device=self.device, dtype=torch.float32)   # mPlaybackMem (float, as the reference)
        self.flags = np.zeros(mem_size, np.int64)
        self.flags_dev = torch.zeros(mem_size, device=self.device, dtype=torch.int64)     # device mirror (no host round trip when a batch is built)
        self._idx_pin = torch.zeros((64, 256), dtype=torch.int64).pin_memory() if self.device.type == "cuda" else None   # ring of staging slots
        self._idx_slot = 0
        self._idx_events = [None] * 64   # one event per staging slot: a slot is rewritten only after its queued copy has executed
        self.rng = np.random.RandomState(seed)
        self.Reset()
        self.UpdateTargetNet()
        # batch-32 steps are launch-latency bound in eager mode: on the GPU the evaluation and the solver step are captured once as
        # HIP graphs (static input / output buffers) and replayed; any capture problem falls back to eager execution
        self.use_graphs = (self.device.type == "cuda") if use_graphs is None else bool(use_graphs)
        self._g_eval, self._g_step = {}, None

    def _configure_sizes(self):
        assert self.A == 1 + self.frag_size and self.desc["in_size"] == self.S
        self.out_size = self.num_frags * (1 + self.frag_size)

    def _new_net(self):
        return MaceNet(self.desc)

    # ---- cNeuralNetTrainer::ResetParams / cMACETrainer::Reset
    def Reset(self):
        self.total_tuples = self.num_tuples = self.head = self.iter = self.actor_iter = 0
        self.stage_train = False
        self.critic_buffer, self.critic_pos = [], {}
        self.actor_buffer, self.actor_pos = [], {}
        self.actor_batch_buffer = []
        self._last_loss = self._last_actor_loss = None

    @property
    def last_loss(self): return None if self._last_loss is None else float(self._last_loss)
    @property
    def last_actor_loss(self): return None if self._last_actor_loss is None else float(self._last_actor_loss)

    # ---- weights / normalisers
    def GetWeights(self): return self.net.get_flat()
    def SetWeights(self, w):
        self.net.set_flat(w); self.UpdateTargetNet()
    def GetOffsetScale(self):
        # host copies, refreshed only after a setter ran: four device read-backs per policy hand-over would each wait behind whatever the GPU is busy with
        c = getattr(self, "_norm_host", None)
        if c is None:
            f = lambda t: t.detach().to(torch.float64).cpu().numpy()
            c = self._norm_host = (f(self.in_off), f(self.in_scale), f(self.out_off), f(self.out_scale))
        return tuple(x.copy() for x in c)
    def SetInputOffsetScale(self, off, scale):   # in place: captured graphs keep pointing at these buffers
        self._norm_host = None
        self.in_off.copy_(torch.as_tensor(off, device=self.device, dtype=self.dtype)); self.in_scale.copy_(torch.as_tensor(scale, device=self.device, dtype=self.dtype))
    def SetOutputOffsetScale(self, off, scale):
        self._norm_host = None
        self.out_off.copy_(torch.as_tensor(off, device=self.device, dtype=self.dtype)); self.out_scale.copy_(torch.as_tensor(scale, device=self.device, dtype=self.dtype))
    def OutputModel(self, model_file):
        """cNeuralNetTrainer::OutputModel -> cNeuralNet::OutputModel (learning/NeuralNet.cpp:1139-1180): the net as a Caffe HDF5 model
        (/data/<layer>/<blob>, Caffe blob shapes) plus '<model>_scale.txt' with the normalisers, both readable by the reference."""
        from . import caffe_hdf5
        names = caffe_hdf5.mace_layer_names(self.num_frags)
        layers = {}
        for name, m in zip(names, self.net.mods):
            w = m.weight.detach().to(torch.float32).cpu().numpy()
            if w.ndim == 3:
                w = w[:, :, None, :]                       # Caffe convolution blob: [out, in, kernel_h = 1, kernel_w]
            layers[name] = [w, m.bias.detach().to(torch.float32).cpu().numpy()]
        caffe_hdf5.write_caffe_model(model_file, layers)
        io, isc, oo, osc = self.GetOffsetScale()
        vec = lambda v: "[" + ", ".join("%f" % x for x in v) + "]"
        with open(os.path.splitext(model_file)[0] + "_scale.txt", "w") as f:
            f.write('{\n"InputOffset": %s,\n"InputScale": %s,\n"OutputOffset": %s,\n"OutputScale": %s\n}' % (vec(io), vec(isc), vec(oo), vec(osc)))

    def LoadModel(self, model_file):
        from . import caffe_hdf5
        self.SetWeights(caffe_hdf5.load_mace_weights(model_file, self.num_frags))

    def GetIter(self): return self.iter
    def GetNumTuples(self): return self.total_tuples      # cNeuralNetTrainer::GetNumTuples returns mTotalTuples (learning/NeuralNetTrainer.cpp:286-289): every tuple ever stored, not the ring's fill
    def EnableTargetNet(self): return self.freeze_target_iters > 0
    def UpdateTargetNet(self):
        """cMACETrainer::UpdateTargetNet; without freezing (freeze_target_iters == 0) the target IS the current net (GetTargetNetID)."""
        self.target.flat.copy_(self.net.flat)

    def _target_net(self):
        return self.target if self.EnableTargetNet() else self.net

    # ---- evaluation in unnormalised space (cNeuralNet::EvalBatch)
    def _eval_eager(self, net, X):
        with torch.no_grad():
            y = net((X.to(self.dtype) + self.in_off) * self.in_scale)
            return y / self.out_scale - self.out_off

    def _eval(self, net, X):
        n = X.shape[0]
        if not self.use_graphs or n > self.batch:
            return self._eval_eager(net, X)
        key = id(net)
        try:
            if key not in self._g_eval:
                xin = torch.zeros((self.batch, self.S), device=self.device, dtype=torch.float32)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._eval_eager(net, xin)
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    yout = self._eval_eager(net, xin)
                self._g_eval[key] = (g, xin, yout)
            g, xin, yout = self._g_eval[key]
            xin[:n].copy_(X)
            g.replay()
            return yout[:n].clone()
        except Exception:   # capture not available: stay eager from now on
            self.use_graphs = False
            return self._eval_eager(net, X)
    def Eval(self, X):
        return self._eval(self.net, torch.as_tensor(np.atleast_2d(X), device=self.device)).to(torch.float64).cpu().numpy()

    # ---- tuples
    def AddTuples(self, rows, flags):
        """rows [n, W] float in the MACE layout, flags [n] tExpTuple bits. Returns the slot of every row (-1 = rejected by CheckTuple)."""
        rows = np.asarray(rows, np.float32).reshape(-1, self.W)
        if rows.shape[0] > self.mem_size:      # more rows than slots: a slot would be written twice by ONE indexed store (order undefined); go ring by ring
            flags = np.asarray(flags, np.int64)
            return np.concatenate([self.AddTuples(rows[k:k + self.mem_size], flags[k:k + self.mem_size]) for k in range(0, rows.shape[0], self.mem_size)])
        ok = np.all(np.isfinite(rows), axis=1)
        slots = np.full(rows.shape[0], -1, np.int64)
        keep = np.nonzero(ok)[0]
        if keep.size:
            slots[keep] = (self.head + np.arange(keep.size)) % self.mem_size
            self._store_rows(slots[keep], rows[keep], np.asarray(flags, np.int64)[keep], keep.size == rows.shape[0])
            for i in keep:   # same order as the reference: write slot, advance head, then UpdateBuffers(slot)
                t = int(slots[i])
                self.flags[t] = int(flags[i])
                self.head = (self.head + 1) % self.mem_size
                self.num_tuples = min(self.mem_size, self.num_tuples + 1)
                self.total_tuples += 1
                self._update_buffers(t)
        return slots

    def _store_rows(self, slots, rows, flags, contiguous):
        """SetTuple for a batch of accepted rows (contiguous: no row of the call was rejected, i.e. row i of the call goes to slots[0] + i modulo the ring)"""
        didx = self._idx(slots)
        self.mem[didx] = torch.as_tensor(rows, device=self.device)
        self.flags_dev[didx] = torch.as_tensor(flags, device=self.device)

    @staticmethod
    def _buf_add(buf, pos, t):
        if t not in pos:
            pos[t] = len(buf); buf.append(t)

    @staticmethod
    def _buf_del(buf, pos, t):
        if t in pos:   # move the last element into the hole (learning/MACETrainer.cpp:752-757, 774-779)
            i = pos.pop(t); last = buf.pop()
            if last != t:
                buf[i] = last; pos[last] = i

    def _update_buffers(self, t):
        exp_actor = bool(self.flags[t] & FLAG_EXP_ACTOR)
        if exp_actor:
            self._buf_add(self.actor_buffer, self.actor_pos, t); self._buf_del(self.critic_buffer, self.critic_pos, t)
        else:
            self._buf_del(self.actor_buffer, self.actor_pos, t); self._buf_add(self.critic_buffer, self.critic_pos, t)
        while t in self.actor_batch_buffer:
            i = self.actor_batch_buffer.index(t); last = self.actor_batch_buffer.pop()
            if i < len(self.actor_batch_buffer):
                self.actor_batch_buffer[i] = last

    # ---- stages
    def UpdateOffsetScale(self):
        """cNeuralNet::CalcOffsetScale over the stored begin states: offset = -mean, scale = 1 / population std (0 where std == 0)."""
        X = self.mem[:self.num_tuples, 1:1 + self.S].to(torch.float64)
        mean = X.mean(0)
        std = ((X - mean) ** 2).mean(0).sqrt()
        scale = torch.where(std == 0, torch.zeros_like(std), 1.0 / std)
        self.SetInputOffsetScale(-mean, scale)

    def Train(self):
        if not self.stage_train and self.num_tuples >= self.num_init_samples and self.num_tuples > 0:
            if self.num_init_samples > 1 and self.init_input_offset_scale:
                self.UpdateOffsetScale()
            self.stage_train = True
        if self.stage_train:
            self.ApplySteps(self.steps_per_iter)

    def ApplySteps(self, n):
        succ = False
        for _ in range(n):
            succ = self.Step()
        if succ:
            self.iter += 1

    def Step(self):
        ids = self.FetchMinibatch(self.batch)
        succ = len(ids) >= self.batch
        if succ:
            X, Y = self._critic_problem(ids)
            self._last_loss = self._solver_step(X, Y)
        self.UpdateActor()
        if self.EnableTargetNet() and self.iter > 0 and self.iter % self.freeze_target_iters == 0:
            self.target.flat.copy_(self.net.flat)
        return succ

    # ---- critic
    def FetchMinibatch(self, size):
        n = len(self.critic_buffer)
        if n < size:
            return []
        buf = self.critic_buffer       # (one array draw = the same stream as `size` scalar draws, at a seventh of the host time: 14 vs 101 us for 32)
        return [buf[i] for i in self.rng.randint(0, n, size=size).tolist()]

    def _idx(self, ids):
        """Index list -> device tensor; on the GPU through page-locked staging so the copy is queued, not synchronous."""
        a = np.asarray(ids, np.int64)
        if self._idx_pin is None or a.size > self._idx_pin.shape[1]:
            return torch.as_tensor(a, device=self.device)
        self._idx_slot = (self._idx_slot + 1) % self._idx_pin.shape[0]
        ev = self._idx_events[self._idx_slot]
        if ev is not None:
            ev.synchronize()   # the copy queued from this slot 64 calls ago has executed (AddTuples in the init stage queues ~1500 copies without a read-back)
        buf = self._idx_pin[self._idx_slot, :a.size]
        buf.copy_(torch.from_numpy(a))
        out = buf.to(self.device, non_blocking=True)
        if ev is None:
            ev = self._idx_events[self._idx_slot] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return out

    def _rows(self, ids):
        return self.mem[self._idx(ids)]

    def _new_q(self, rows, idx):
        """CalcNewCumulativeRewardBatch: r (1 - discount) + discount max_frag Q_target(s'), or r (1 - discount) on failure."""
        r = rows[:, 0].to(self.dtype) * (1.0 - self.discount)
        q_end = self._eval(self._target_net(), rows[:, 1 + self.S + self.A:])[:, :self.num_frags].max(1).values
        fail = (self.flags_dev[idx] & FLAG_FAIL) != 0
        return torch.where(fail, r, r + self.discount * q_end)

    def _critic_problem(self, ids):
        idx = self._idx(ids)
        rows = self.mem[idx]
        X = rows[:, 1:1 + self.S]
        Y = self._eval(self.net, X)
        a = rows[:, 1 + self.S].to(torch.int64)
        Y.scatter_(1, a[:, None], self._new_q(rows, idx)[:, None].to(Y.dtype))
        return X, Y

    # ---- actor
    def FetchActorMinibatch(self, size):
        n = len(self.actor_buffer)
        out = []
        if n == 0:
            return out
        taken = set(self.actor_batch_buffer)
        buf = self.actor_buffer
        for i in self.rng.randint(0, n, size=min(size, n)).tolist():     # every draw happens whether or not its slot is accepted: one array draw, same stream
            t = buf[i]
            if t not in taken:
                taken.add(t); out.append(t)
        return out

    def UpdateActorBatchBuffer(self):
        ids = self.FetchActorMinibatch(self.batch)
        if not ids:
            return
        idx = self._idx(ids)
        rows = self.mem[idx]
        curr = self._eval(self._target_net(), rows[:, 1:1 + self.S])[:, :self.num_frags].max(1).values
        new = self._new_q(rows, idx)
        better = (new > curr).cpu().numpy()
        self.actor_batch_buffer += [t for t, b in zip(ids, better) if b]

    def UpdateActor(self):
        if self.stage_train:
            self.UpdateActorBatchBuffer()
        for _ in range(len(self.actor_batch_buffer) // self.batch):
            ids = self.actor_batch_buffer[:self.batch]
            rows = self._rows(ids)
            X = rows[:, 1:1 + self.S]
            Y = self._eval(self.net, X)
            a = rows[:, 1 + self.S].to(torch.int64)
            frag = rows[:, 2 + self.S:1 + self.S + self.A].to(self.dtype)
            cols = self.num_frags + a[:, None] * self.frag_size + torch.arange(self.frag_size, device=self.device)[None, :]
            Y.scatter_(1, cols, frag)
            self._last_actor_loss = self._solver_step(X, Y)
            self.actor_iter += 1
            del self.actor_batch_buffer[:self.batch]

    # ---- cNeuralNet::Train -> one Caffe SGD iteration on one batch
    def _lr(self):
        s = self.solver
        if s["lr_policy"] == "step":
            return s["base_lr"] * s["gamma"] ** int(self.solver_iter // s["stepsize"])
        if s["lr_policy"] == "inv":
            return s["base_lr"] * (1 + s["gamma"] * self.solver_iter) ** (-s["power"])
        return s["base_lr"]

    solver_iter = 0

    def _solver_step_body(self, X, Y):
        x = (X.to(self.dtype) + self.in_off) * self.in_scale                  # LoadTrainData: data and labels go in normalised
        label = (Y + self.out_off) * self.out_scale
        self.net.gflat.zero_()
        out = self.net(x)
        loss = 0.5 * ((out - label) ** 2).sum() / x.shape[0]                   # EuclideanLoss
        loss.backward()
        rate, mom, wd = self._lr(), self.solver["momentum"], self.solver["weight_decay"]
        with torch.no_grad():
            diff = torch.addcmul(self.net.gflat, self.decay_mult, self.net.flat, value=wd)   # Regularize (L2): diff + wd * decay_mult * w
            self.hflat.mul_(mom).addcmul_(self.rate_mult, diff, value=rate)                    # ComputeUpdateValue
            self.net.flat.sub_(self.hflat)                                                     # Net::Update
        return loss.detach()

    def _solver_step(self, X, Y):
        if self.use_graphs and self.solver["lr_policy"] == "fixed" and X.shape[0] == self.batch:
            try:
                if self._g_step is None:
                    xs = torch.zeros((self.batch, self.S), device=self.device, dtype=torch.float32)
                    ys = torch.zeros((self.batch, self.out_size), device=self.device, dtype=self.dtype)
                    keep = (self.net.flat.clone(), self.hflat.clone())
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(2):
                            self._solver_step_body(xs, ys)
                    torch.cuda.current_stream().wait_stream(side)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        lo = self._solver_step_body(xs, ys)
                    self.net.flat.copy_(keep[0]); self.hflat.copy_(keep[1])     # undo the warm-up / capture updates
                    self._g_step = (g, xs, ys, lo)
                g, xs, ys, lo = self._g_step
                xs.copy_(X); ys.copy_(Y)
                g.replay()
                self.solver_iter += 1
                return lo.clone()
            except Exception:
                self.use_graphs = False; self._g_step = None
        loss = self._solver_step_body(X, Y)
        self.solver_iter += 1
        return loss


class QNetTrainer(MACETrainer):
    """cQNetTrainer (learning/QNetTrainer.cpp; pool size 1, synchronous mode) for the Q head (-char_ctrl= dog / raptor, args/opt_args_train_q.txt): replay
    rows [r | s | one-hot a | s'] exactly as the rollout engine emits them, uniform minibatches over the whole replay memory
    (cNeuralNetTrainer::FetchMinibatch, learning/NeuralNetTrainer.cpp:508-524), targets y[a] = r (1 - discount) on failure, else
    r (1 - discount) + discount * Q_ref(s')[argmax_a' Q(s')] with the reference net = the net itself for a pool of one (GetRandRefID, :183-192;
    FREEZE_TARGET_NET is compiled out in the reference), Caffe SGD as in MACETrainer. Everything stays on the device."""

    def _configure_sizes(self):
        assert self.num_frags == 0 and self.A == self.frag_size and self.desc["in_size"] == self.S, "the Q net has one output per base action"
        self.out_size = self.frag_size

    def _new_net(self):
        return QNet(self.desc)

    def _update_buffers(self, t):
        pass                                                            # no critic / actor index buffers: every stored tuple is a candidate

    def EnableTargetNet(self):
        return False

    def FetchMinibatch(self, size):
        n = self.num_tuples
        return self.rng.randint(0, n, size=size).tolist() if n > 0 else []

    def _q_problem(self, ids):
        idx = self._idx(ids)
        rows = self.mem[idx]
        X = rows[:, 1:1 + self.S]
        Y = self._eval(self.net, X)
        y_next = self._eval(self.net, rows[:, 1 + self.S + self.A:])     # curr_net and ref_net coincide for a pool of one
        a_next = y_next.argmax(1, keepdim=True)
        q_end = y_next.gather(1, a_next)[:, 0]
        r = rows[:, 0].to(self.dtype) * (1.0 - self.discount)
        fail = (self.flags_dev[idx] & FLAG_FAIL) != 0
        new_q = torch.where(fail, r, r + self.discount * q_end)
        a = rows[:, 1 + self.S:1 + self.S + self.A].argmax(1, keepdim=True)      # tuple.mAction.maxCoeff(&action_idx): the first maximum of the one-hot
        Y.scatter_(1, a, new_q[:, None].to(Y.dtype))
        return X, Y

    def Step(self):
        ids = self.FetchMinibatch(self.batch)
        if len(ids) >= self.batch:
            X, Y = self._q_problem(ids)
            self._last_loss = self._solver_step(X, Y)
        return True                                                     # cQNetTrainer::Step returns true unconditionally (learning/QNetTrainer.cpp:142-163)

    def UpdateActor(self):
        pass

    def OutputModel(self, model_file):
        raise NotImplementedError("HDF5 layer names of the Q net are not mapped; use GetWeights() / BatchScenario.WriteOffsetScale")


class CaclaTrainer(QNetTrainer):
    """cCaclaTrainer on top of cACTrainer (learning/CaclaTrainer.cpp, learning/ACTrainer.cpp; pool of one, synchronous, mode eModeCacla, reward mode
    "start") for -char_ctrl= dog_cacla (args/opt_args_train_cacla.txt): a critic V(s) (dog_critic_*.prototxt, one output) and an actor
    (dog_actor_*.prototxt, the optimisable action parameters), both single-head nets, one replay ring of rows [r | s | a | s'] with the flags
    fail (1) / off-policy (2) as the rollout engine emits them.
      critic step   uniform minibatch, target r (1 - g) on failure else r (1 - g) + g V_target(s')            (CaclaTrainer.cpp:149-157, 234-277)
      actor step    candidates drawn from the OFF-POLICY tuples (exploration steps), kept when the critic's TD error
                    new_v - V_target(s) is positive; per full actor batch one SGD step of the actor towards the explored
                    actions themselves (CACLA)                                                            (:105-127, 342-387; ACTrainer.cpp:611-646, 271-284)
      target net    refreshed every trainer_freeze_target_iters critic iterations when that is > 0, else the critic itself (:395-418)
    GetWeights / GetOffsetScale / SetOutputOffsetScale speak for the ACTOR (what the rollout engine runs); the critic has its own accessors."""

    def __init__(self, critic_net_file, critic_solver_file, actor_net_file, actor_solver_file, state_size, action_size, **kw):
        actor_kw = dict(kw); actor_kw.update(mem_size=1, num_init_samples=1)
        self._actor_args = (actor_net_file, actor_solver_file, state_size, action_size, actor_kw)
        super().__init__(critic_net_file, critic_solver_file, state_size, action_size, **kw)
        # cBaseControllerCacla::BuildCriticOutputOffsetScale (sim/BaseControllerCacla.cpp:197-202)
        MACETrainer.SetOutputOffsetScale(self, np.full(1, -0.5), np.full(1, 2.0))

    def _configure_sizes(self):
        assert self.num_frags == 0 and self.frag_size == 1 and self.desc["in_size"] == self.S, "the critic has one output"
        self.out_size = 1

    def Reset(self):
        super().Reset()
        net_file, solver_file, S, A, kw = self._actor_args
        self.actor = QNetTrainer(net_file, solver_file, S, A, **kw)          # the actor net with its own solver state, normalisers and captured graphs
        self.actor_batch = self.actor.batch
        self.off_policy_buffer, self.off_policy_pos = [], {}
        self.actor_batch_buffer, self.actor_batch_td = [], []

    # ---- the interface the training loop uses: the policy = the actor
    def GetWeights(self): return self.actor.GetWeights()
    def SetWeights(self, w): self.actor.SetWeights(w)
    def GetOffsetScale(self): return self.actor.GetOffsetScale()
    def SetOutputOffsetScale(self, off, scale): self.actor.SetOutputOffsetScale(off, scale)
    def SetInputOffsetScale(self, off, scale):          # cACTrainer::SetInputOffsetScale: both nets
        MACETrainer.SetInputOffsetScale(self, off, scale); self.actor.SetInputOffsetScale(off, scale)
    def GetCriticWeights(self): return self.net.get_flat()
    def SetCriticWeights(self, w):
        self.net.set_flat(w); self.UpdateTargetNet()
    def GetCriticOffsetScale(self): return MACETrainer.GetOffsetScale(self)
    def EnableTargetNet(self): return self.freeze_target_iters > 0
    @property
    def last_actor_loss(self): return self.actor.last_loss

    def _update_buffers(self, t):
        if self.flags[t] & FLAG_OFF_POLICY:
            self._buf_add(self.off_policy_buffer, self.off_policy_pos, t)
        else:
            self._buf_del(self.off_policy_buffer, self.off_policy_pos, t)
        while t in self.actor_batch_buffer:              # a slot that is overwritten leaves the pending actor batch (move-last-into-hole, CaclaTrainer.cpp:445-460)
            i = self.actor_batch_buffer.index(t)
            last, last_td = self.actor_batch_buffer.pop(), self.actor_batch_td.pop()
            if i < len(self.actor_batch_buffer):
                self.actor_batch_buffer[i] = last; self.actor_batch_td[i] = last_td

    def _new_v(self, rows, idx):
        r = rows[:, 0].to(self.dtype) * (1.0 - self.discount)                 # NormalizeReward, eRewardModeStart
        v_end = self._eval(self._target_net(), rows[:, 1 + self.S + self.A:])[:, 0]
        fail = (self.flags_dev[idx] & FLAG_FAIL) != 0
        return torch.where(fail, r, r + self.discount * v_end)

    def Step(self):
        ids = self.FetchMinibatch(self.batch)
        if len(ids) >= self.batch:
            idx = self._idx(ids)
            rows = self.mem[idx]
            self._last_loss = self._solver_step(rows[:, 1:1 + self.S], self._new_v(rows, idx)[:, None].to(self.dtype))
        self.UpdateActor()
        if self.EnableTargetNet() and self.iter > 0 and self.iter % self.freeze_target_iters == 0:      # cCaclaTrainer::Step: CheckUpdateTarget after cACTrainer::Step
            self.UpdateTargetNet()
        return True

    def _draw_actor_candidates(self):
        n = len(self.off_policy_buffer)
        ids = []
        if n == 0:
            return ids
        taken = set(self.actor_batch_buffer)
        buf = self.off_policy_buffer
        for i in self.rng.randint(0, n, size=min(self.actor_batch, n)).tolist():
            t = buf[i]
            if t not in taken:
                taken.add(t); ids.append(t)
        return ids

    def UpdateActorBatchBuffer(self):
        ids = self._draw_actor_candidates()
        if not ids:
            return
        idx = self._idx(ids)
        rows = self.mem[idx]
        curr = self._eval(self._target_net(), rows[:, 1:1 + self.S])[:, 0]
        td = (self._new_v(rows, idx) - curr).to(torch.float64).cpu().numpy()
        for t, d in zip(ids, td):
            if d > 0:
                self.actor_batch_buffer.append(t); self.actor_batch_td.append(float(d))

    def UpdateActor(self):
        if self.stage_train:
            self.UpdateActorBatchBuffer()
        for _ in range(len(self.actor_batch_buffer) // self.actor_batch):
            rows = self._rows(self.actor_batch_buffer[:self.actor_batch])
            X = rows[:, 1:1 + self.S]
            Y = rows[:, 1 + self.S:1 + self.S + self.A].to(self.dtype)          # BuildTupleActorY: the action that was taken
            self.actor._last_loss = self.actor._solver_step(X, Y)
            self.actor_iter += 1
            del self.actor_batch_buffer[:self.actor_batch]; del self.actor_batch_td[:self.actor_batch]

    def UpdateOffsetScale(self):
        super().UpdateOffsetScale()                                            # UpdateCriticOffsetScale ...
        io, isc, _, _ = MACETrainer.GetOffsetScale(self)
        self.actor.SetInputOffsetScale(io, isc)                                 # ... and UpdateActorOffsetScale: the same statistics of the same states


def anneal(it, n_iters, v0, v1):
    """cScenarioTrain::CalcExpRate / CalcExpTemp / CalcExpBaseRate: linear interpolation over the anneal window."""
    lerp = min(max(float(it) / n_iters, 0.0), 1.0) if n_iters > 0 else 1.0
    return (1 - lerp) * v0 + lerp * v1
