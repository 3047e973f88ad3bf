"""ORACLE (test infrastructure, NOT product code): plain-Python / numpy fp64 restatement of the MACE trainer's bookkeeping.

Follows, line by line, the reference's
  replay ring + CheckTuple             learning/NeuralNetTrainer.cpp:145-165, 541-576
  critic / actor index buffers         learning/MACETrainer.cpp:731-799 (UpdateBuffers; remove = move last element into the hole)
  critic target                        learning/MACETrainer.cpp:478-515 (CalcNewCumulativeRewardBatch), :226-250 (BuildProblemY)
  actor candidate filter + labels      learning/MACETrainer.cpp:577-600 (UpdateActorBatchBuffer), :285-305 (BuildActorProblemY)
  input normaliser                     learning/NeuralNet.cpp:280-313 (CalcOffsetScale)
  Caffe SGD step                       SGDSolver::ComputeUpdateValue of the pinned Caffe fork (README.md:15): L2 regularise with
                                       weight_decay * decay_mult, history = momentum * history + base_lr * lr_mult * diff, w -= history
The network itself is passed in as a callable (the tests use a numpy forward of a tiny dense net with hand-derived gradients),
so nothing here depends on torch. Parity unpinned beyond these formulas: the reference ships no trainer tests or trained blobs.
"""
import numpy as np


class RefTrainerBook:
    def __init__(self, S, A, num_frags, frag_size, mem_size, batch, discount):
        self.S, self.A, self.nf, self.fs, self.mem_size, self.batch, self.discount = S, A, num_frags, frag_size, mem_size, batch, discount
        self.W = 1 + 2 * S + A
        self.mem = np.zeros((mem_size, self.W), np.float32)
        self.flags = np.zeros(mem_size, np.int64)
        self.head = self.num = 0
        self.critic, self.actor, self.actor_batch = [], [], []

    @staticmethod
    def _remove(buf, t):
        if t in buf:
            i = buf.index(t); last = buf[-1]; buf[i] = last; buf.pop()

    def add(self, row, flag):
        if not np.all(np.isfinite(row)):
            return -1
        t = self.head
        self.mem[t] = row; self.flags[t] = flag
        self.head = (self.head + 1) % self.mem_size
        self.num = min(self.mem_size, self.num + 1)
        exp_actor = bool(flag & 4)
        if exp_actor:
            if t not in self.actor: self.actor.append(t)
            self._remove(self.critic, t)
        else:
            self._remove(self.actor, t)
            if t not in self.critic: self.critic.append(t)
        while t in self.actor_batch:
            self._remove(self.actor_batch, t)
        return t

    def new_q(self, t, target_eval):
        row = self.mem[t].astype(np.float64)
        r = row[0] * (1.0 - self.discount)
        if self.flags[t] & 1:
            return r
        y = target_eval(row[1 + self.S + self.A:])
        return r + self.discount * np.max(y[:self.nf])

    def critic_label(self, t, net_eval, target_eval):
        row = self.mem[t].astype(np.float64)
        y = net_eval(row[1:1 + self.S]).copy()
        y[int(row[1 + self.S])] = self.new_q(t, target_eval)
        return y

    def actor_label(self, t, net_eval):
        row = self.mem[t].astype(np.float64)
        y = net_eval(row[1:1 + self.S]).copy()
        a = int(row[1 + self.S])
        y[self.nf + a * self.fs: self.nf + (a + 1) * self.fs] = row[2 + self.S:1 + self.S + self.A]
        return y

    def actor_accepts(self, t, target_eval):
        row = self.mem[t].astype(np.float64)
        curr = np.max(target_eval(row[1:1 + self.S])[:self.nf])
        return self.new_q(t, target_eval) > curr

    def offset_scale(self):
        X = self.mem[:self.num, 1:1 + self.S].astype(np.float64)
        n = X.shape[0]
        off = np.zeros(self.S)
        for i in range(n): off += X[i] / n
        sc = np.zeros(self.S)
        for i in range(n): sc += (X[i] - off) ** 2 / n
        sc = np.sqrt(sc)
        return -off, np.where(sc == 0, 0.0, 1.0 / np.where(sc == 0, 1.0, sc))


def caffe_sgd_step(w, grad, hist, base_lr, momentum, weight_decay, lr_mult, decay_mult):
    diff = grad + weight_decay * decay_mult * w
    hist_new = momentum * hist + base_lr * lr_mult * diff
    return w - hist_new, hist_new


# ---------------------------------------------------------------------------------------------------------------------------------
# A whole trainer in numpy fp64: the net (forward + hand-derived backward), the solver and the iteration order of cMACETrainer.
# ---------------------------------------------------------------------------------------------------------------------------------
class RefMaceNet:
    """The *_mace3 topology (3 valid 1-D convolutions over the terrain slice -> terr_ip0 -> concat with the character slice -> ip0 -> value head
    val_ip0/val_ip1 + n_frags actor heads a{f}_ip0/a{f}_ip1; ReLU after every layer but the head outputs), Caffe conventions: cross-correlation,
    channel-major flatten, blob order conv0..2, terr_ip0, ip0, val_ip0, val_ip1, a{f}_ip0, a{f}_ip1, weight then bias. All arithmetic fp64."""

    def __init__(self, n_terrain, n_char, convs, fc_terr, fc_trunk, fc_head, n_frags, frag_size):
        self.n_terrain, self.n_char, self.n_frags, self.frag_size = n_terrain, n_char, n_frags, frag_size
        self.shapes = []
        cin, w = 1, n_terrain
        for cout, k in convs:
            self.shapes += [(cout, cin, k), (cout,)]; cin, w = cout, w - k + 1
        self.conv_out = (cin, w)
        for nout, nin in [(fc_terr, cin * w), (fc_trunk, fc_terr + n_char), (fc_head, fc_trunk), (n_frags, fc_head)]:
            self.shapes += [(nout, nin), (nout,)]
        for _ in range(n_frags):
            self.shapes += [(fc_head, fc_trunk), (fc_head,), (frag_size, fc_head), (frag_size,)]
        self.sizes = [int(np.prod(s)) for s in self.shapes]
        self.num_params = sum(self.sizes)

    def split(self, flat):
        out, off = [], 0
        for s, n in zip(self.shapes, self.sizes):
            out.append(np.asarray(flat[off:off + n], np.float64).reshape(s)); off += n
        assert off == len(flat)
        return out

    @staticmethod
    def _im2col(x, k):   # x [B, cin, W] -> [B, Wout, cin * k] (column index = channel * k + tap)
        B, cin, W = x.shape
        wout = W - k + 1
        cols = np.empty((B, wout, cin, k))
        for t in range(k):
            cols[:, :, :, t] = x[:, :, t:t + wout].transpose(0, 2, 1)
        return cols.reshape(B, wout, cin * k)

    def forward(self, flat, x, keep=False):
        P = self.split(flat)
        B = x.shape[0]
        t = x[:, :self.n_terrain].reshape(B, 1, self.n_terrain)
        tape = []
        for l in range(3):
            Wc, bc = P[2 * l], P[2 * l + 1]
            cols = self._im2col(t, Wc.shape[2])
            z = cols @ Wc.reshape(Wc.shape[0], -1).T + bc          # [B, Wout, cout]
            tape.append((cols, z, t.shape))
            t = np.maximum(z, 0).transpose(0, 2, 1)
        f_in = t.reshape(B, -1)
        z3 = f_in @ P[6].T + P[7]; a3 = np.maximum(z3, 0)
        c_in = np.concatenate([a3, x[:, self.n_terrain:]], 1)
        z4 = c_in @ P[8].T + P[9]; h = np.maximum(z4, 0)
        heads, outs = [], []
        zv = h @ P[10].T + P[11]; av = np.maximum(zv, 0); outs.append(av @ P[12].T + P[13]); heads.append((zv, av))
        for f in range(self.n_frags):
            b0 = 14 + 4 * f
            zf = h @ P[b0].T + P[b0 + 1]; af = np.maximum(zf, 0); outs.append(af @ P[b0 + 2].T + P[b0 + 3]); heads.append((zf, af))
        y = np.concatenate(outs, 1)
        if keep:
            self._tape = (P, x, tape, f_in, z3, c_in, z4, h, heads)
        return y

    def backward(self, dy):
        """Gradient of sum(dy * y) wrt every blob, flat, in blob order (call after forward(..., keep=True))."""
        P, x, tape, f_in, z3, c_in, z4, h, heads = self._tape
        G = [None] * len(P)
        B = dy.shape[0]
        dh = np.zeros_like(h)
        col = 0
        for i, (zf, af) in enumerate(heads):
            b0 = 10 if i == 0 else 14 + 4 * (i - 1)
            nout = P[b0 + 2].shape[0]
            d_out = dy[:, col:col + nout]; col += nout
            G[b0 + 2] = d_out.T @ af; G[b0 + 3] = d_out.sum(0)
            dz = (d_out @ P[b0 + 2]) * (zf > 0)
            G[b0] = dz.T @ h; G[b0 + 1] = dz.sum(0)
            dh += dz @ P[b0]
        dz4 = dh * (z4 > 0)
        G[8] = dz4.T @ c_in; G[9] = dz4.sum(0)
        da3 = (dz4 @ P[8])[:, :z3.shape[1]]
        dz3 = da3 * (z3 > 0)
        G[6] = dz3.T @ f_in; G[7] = dz3.sum(0)
        dt = (dz3 @ P[6]).reshape(B, *self.conv_out)                 # [B, cout, Wout]
        for l in (2, 1, 0):
            cols, z, in_shape = tape[l]
            Wc = P[2 * l]
            dz = dt.transpose(0, 2, 1) * (z > 0)                     # [B, Wout, cout]
            G[2 * l] = np.einsum("bwo,bwc->oc", dz, cols).reshape(Wc.shape); G[2 * l + 1] = dz.sum((0, 1))
            if l > 0:
                dcols = (dz @ Wc.reshape(Wc.shape[0], -1)).reshape(B, dz.shape[1], in_shape[1], Wc.shape[2])
                dt = np.zeros(in_shape)
                for t in range(Wc.shape[2]):
                    dt[:, :, t:t + dz.shape[1]] += dcols[:, :, :, t].transpose(0, 2, 1)
        return np.concatenate([g.reshape(-1) for g in G])


class RefMaceTrainer:
    """cMACETrainer, pool size 1, synchronous mode, fp64 numpy: the iteration order of cNeuralNetTrainer::Train / ApplySteps / Step
    (learning/NeuralNetTrainer.cpp:96-141, 392-455) and cMACETrainer::Step / UpdateActor (learning/MACETrainer.cpp:252-283, 517-600) on top of the
    bookkeeping above. Index draws come from numpy's RandomState(seed).randint (the reference draws from its clock-seeded cRand): the product
    trainer is run with the same stream so that batches coincide."""

    def __init__(self, net, blob_mults, S, A, mem_size, batch, discount, num_init_samples, solver, seed, freeze_target_iters=0, init_input_offset_scale=True):
        self.net, self.S, self.A, self.batch, self.discount = net, S, A, batch, discount
        self.nf, self.fs = net.n_frags, net.frag_size
        self.book = RefTrainerBook(S, A, self.nf, self.fs, mem_size, batch, discount)
        self.num_init_samples, self.solver, self.freeze, self.init_os = num_init_samples, solver, freeze_target_iters, init_input_offset_scale
        self.lr_mult = np.concatenate([np.full(n, m[0]) for n, m in zip(net.sizes, blob_mults)])
        self.decay_mult = np.concatenate([np.full(n, m[1]) for n, m in zip(net.sizes, blob_mults)])
        self.w = np.zeros(net.num_params); self.w_target = self.w.copy(); self.hist = np.zeros(net.num_params)
        out = self.nf * (1 + self.fs)
        self.in_off, self.in_scale, self.out_off, self.out_scale = np.zeros(S), np.ones(S), np.zeros(out), np.ones(out)
        self.rng = np.random.RandomState(seed)
        self.iter = self.actor_iter = 0
        self.stage_train = False
        self.last_loss = None

    def set_weights(self, w):
        self.w = np.asarray(w, np.float64).copy(); self.w_target = self.w.copy()

    def eval(self, w, X):
        X = np.atleast_2d(np.asarray(X, np.float64))
        return self.net.forward(w, (X + self.in_off) * self.in_scale) / self.out_scale - self.out_off

    def _target(self):
        return self.w_target if self.freeze > 0 else self.w

    def add_tuples(self, rows, flags):
        return [self.book.add(np.asarray(r, np.float32), int(f)) for r, f in zip(rows, flags)]

    def _sgd(self, X, Y):
        x = (np.asarray(X, np.float64) + self.in_off) * self.in_scale
        label = (Y + self.out_off) * self.out_scale
        out = self.net.forward(self.w, x, keep=True)
        loss = 0.5 * ((out - label) ** 2).sum() / x.shape[0]
        grad = self.net.backward((out - label) / x.shape[0])
        s = self.solver
        self.w, self.hist = caffe_sgd_step(self.w, grad, self.hist, s["base_lr"], s["momentum"], s["weight_decay"], self.lr_mult, self.decay_mult)
        return loss

    def train(self):
        b = self.book
        if not self.stage_train and b.num >= self.num_init_samples and b.num > 0:
            if self.num_init_samples > 1 and self.init_os:
                self.in_off, self.in_scale = b.offset_scale()
            self.stage_train = True
        if self.stage_train and self.step():
            self.iter += 1

    def step(self):
        b = self.book
        n = len(b.critic)
        ids = [b.critic[int(self.rng.randint(0, n))] for _ in range(self.batch)] if n >= self.batch else []
        succ = len(ids) >= self.batch
        net_eval = lambda x: self.eval(self.w, x)[0]
        tgt_eval = lambda x: self.eval(self._target(), x)[0]
        if succ:
            X = np.stack([b.mem[t, 1:1 + self.S] for t in ids]).astype(np.float64)
            Y = np.stack([b.critic_label(t, net_eval, tgt_eval) for t in ids])
            self.last_loss = self._sgd(X, Y)
            net_eval = lambda x: self.eval(self.w, x)[0]
        # UpdateActor
        if self.stage_train:
            na = len(b.actor)
            drawn = []
            for _ in range(min(self.batch, na)):
                t = b.actor[int(self.rng.randint(0, na))]
                if t not in b.actor_batch and t not in drawn:
                    drawn.append(t)
            b.actor_batch += [t for t in drawn if b.actor_accepts(t, tgt_eval)]
        for _ in range(len(b.actor_batch) // self.batch):
            ids = b.actor_batch[:self.batch]
            X = np.stack([b.mem[t, 1:1 + self.S] for t in ids]).astype(np.float64)
            Y = np.stack([b.actor_label(t, net_eval) for t in ids])
            self._sgd(X, Y)
            net_eval = lambda x: self.eval(self.w, x)[0]
            self.actor_iter += 1
            del b.actor_batch[:self.batch]
        if self.freeze > 0 and self.iter > 0 and self.iter % self.freeze == 0:
            self.w_target = self.w.copy()
        return succ


# ---------------------------------------------------------------------------------------------------------------------------------
# The Q head's trainer (learning/QNetTrainer.cpp) in numpy fp64.
# ---------------------------------------------------------------------------------------------------------------------------------
class RefQNet:
    """Single-head net of data/policies/dog/nets/dog_q_*.prototxt: the MACE trunk (3 valid conv1d -> terr_ip0 -> concat) -> ip1 -> ip2 -> output,
    ReLU after every layer but the output. Blob order conv0..2, terr_ip0, ip1, ip2, output; weight then bias."""

    def __init__(self, n_terrain, n_char, convs, fc_terr, fc1, fc2, n_out):
        self.n_terrain, self.n_char = n_terrain, n_char
        self.shapes = []
        cin, w = 1, n_terrain
        for cout, k in convs:
            self.shapes += [(cout, cin, k), (cout,)]; cin, w = cout, w - k + 1
        self.conv_out = (cin, w)
        for nout, nin in [(fc_terr, cin * w), (fc1, fc_terr + n_char), (fc2, fc1), (n_out, fc2)]:
            self.shapes += [(nout, nin), (nout,)]
        self.sizes = [int(np.prod(s)) for s in self.shapes]
        self.num_params = sum(self.sizes)

    split = RefMaceNet.split
    _im2col = staticmethod(RefMaceNet._im2col)

    def forward(self, flat, x, keep=False):
        P = self.split(flat)
        B = x.shape[0]
        t = x[:, :self.n_terrain].reshape(B, 1, self.n_terrain)
        tape = []
        for l in range(3):
            Wc, bc = P[2 * l], P[2 * l + 1]
            cols = self._im2col(t, Wc.shape[2])
            z = cols @ Wc.reshape(Wc.shape[0], -1).T + bc
            tape.append((cols, z, t.shape))
            t = np.maximum(z, 0).transpose(0, 2, 1)
        acts = [t.reshape(B, -1)]; zs = []
        for i, b0 in enumerate((6, 8, 10, 12)):
            a_in = acts[-1] if i != 1 else np.concatenate([acts[-1], x[:, self.n_terrain:]], 1)
            if i == 1: acts[-1] = a_in
            z = a_in @ P[b0].T + P[b0 + 1]; zs.append(z)
            acts.append(np.maximum(z, 0) if i < 3 else z)
        if keep:
            self._tape = (P, tape, acts, zs)
        return acts[-1]

    def backward(self, dy):
        P, tape, acts, zs = self._tape
        G = [None] * len(P)
        B = dy.shape[0]
        d = dy
        for i, b0 in reversed(list(enumerate((6, 8, 10, 12)))):
            if i < 3: d = d * (zs[i] > 0)
            G[b0] = d.T @ acts[i]; G[b0 + 1] = d.sum(0)
            d = d @ P[b0]
            if i == 1: d = d[:, :zs[0].shape[1]]
        dt = d.reshape(B, *self.conv_out)
        for l in (2, 1, 0):
            cols, z, in_shape = tape[l]
            Wc = P[2 * l]
            dz = dt.transpose(0, 2, 1) * (z > 0)
            G[2 * l] = np.einsum("bwo,bwc->oc", dz, cols).reshape(Wc.shape); G[2 * l + 1] = dz.sum((0, 1))
            if l > 0:
                dcols = (dz @ Wc.reshape(Wc.shape[0], -1)).reshape(B, dz.shape[1], in_shape[1], Wc.shape[2])
                dt = np.zeros(in_shape)
                for t in range(Wc.shape[2]):
                    dt[:, :, t:t + dz.shape[1]] += dcols[:, :, :, t].transpose(0, 2, 1)
        return np.concatenate([g.reshape(-1) for g in G])


class RefQTrainer:
    """cQNetTrainer with a pool of one (learning/QNetTrainer.cpp:27-83 BuildProblemY, :142-163 Step; cNeuralNetTrainer::AddTuple / FetchMinibatch /
    Train, learning/NeuralNetTrainer.cpp:145-193, 508-524): ring of rows [r | s | one-hot a | s'], uniform minibatches over the stored tuples,
    y[a] = r (1 - g) on failure else r (1 - g) + g Q(s')[argmax Q(s')] (reference net = the net itself), Caffe SGD."""

    def __init__(self, net, blob_mults, S, A, mem_size, batch, discount, num_init_samples, solver, seed, init_input_offset_scale=True):
        self.net, self.S, self.A, self.batch, self.discount = net, S, A, batch, discount
        self.W = 1 + 2 * S + A
        self.mem = np.zeros((mem_size, self.W), np.float32); self.flags = np.zeros(mem_size, np.int64)
        self.mem_size, self.head, self.num = mem_size, 0, 0
        self.num_init_samples, self.solver, self.init_os = num_init_samples, solver, init_input_offset_scale
        self.lr_mult = np.concatenate([np.full(n, m[0]) for n, m in zip(net.sizes, blob_mults)])
        self.decay_mult = np.concatenate([np.full(n, m[1]) for n, m in zip(net.sizes, blob_mults)])
        self.w = np.zeros(net.num_params); self.hist = np.zeros(net.num_params)
        self.in_off, self.in_scale, self.out_off, self.out_scale = np.zeros(S), np.ones(S), np.zeros(A), np.ones(A)
        self.rng = np.random.RandomState(seed)
        self.iter = 0; self.stage_train = False; self.last_loss = None

    def add_tuples(self, rows, flags):
        for r, f in zip(rows, flags):
            r = np.asarray(r, np.float32)
            if not np.all(np.isfinite(r)):
                continue
            self.mem[self.head] = r; self.flags[self.head] = int(f)
            self.head = (self.head + 1) % self.mem_size; self.num = min(self.mem_size, self.num + 1)

    def eval(self, X):
        X = np.atleast_2d(np.asarray(X, np.float64))
        return self.net.forward(self.w, (X + self.in_off) * self.in_scale) / self.out_scale - self.out_off

    def train(self):
        if not self.stage_train and self.num >= self.num_init_samples and self.num > 0:
            if self.num_init_samples > 1 and self.init_os:
                X = self.mem[:self.num, 1:1 + self.S].astype(np.float64)
                mean = X.mean(0); std = np.sqrt(((X - mean) ** 2).mean(0))
                self.in_off, self.in_scale = -mean, np.where(std == 0, 0.0, 1.0 / np.where(std == 0, 1.0, std))
            self.stage_train = True
        if not self.stage_train:
            return
        ids = [int(self.rng.randint(0, self.num)) for _ in range(self.batch)]
        S, A, g = self.S, self.A, self.discount
        X = self.mem[ids, 1:1 + S].astype(np.float64)
        Y = self.eval(X).copy()
        y_next = self.eval(self.mem[ids, 1 + S + A:].astype(np.float64))
        for i, t in enumerate(ids):
            r = float(self.mem[t, 0]) * (1.0 - g)
            a = int(np.argmax(self.mem[t, 1 + S:1 + S + A]))
            Y[i, a] = r if (self.flags[t] & 1) else r + g * y_next[i, int(np.argmax(y_next[i]))]
        x = (X + self.in_off) * self.in_scale
        label = (Y + self.out_off) * self.out_scale
        out = self.net.forward(self.w, x, keep=True)
        self.last_loss = 0.5 * ((out - label) ** 2).sum() / x.shape[0]
        grad = self.net.backward((out - label) / x.shape[0])
        s = self.solver
        self.w, self.hist = caffe_sgd_step(self.w, grad, self.hist, s["base_lr"], s["momentum"], s["weight_decay"], self.lr_mult, self.decay_mult)
        self.iter += 1


class RefCaclaTrainer:
    """cCaclaTrainer / cACTrainer (pool of one, eModeCacla, reward mode "start") in numpy fp64: critic and actor are RefQNet instances.
    learning/CaclaTrainer.cpp:105-127 FetchActorMinibatch, :137-157 Step / BuildProblemY, :234-277 CalcNewCumulativeRewardBatch, :342-387
    UpdateActorBatchBuffer, :420-462 UpdateBuffers; learning/ACTrainer.cpp:397-412 Step, :611-646 UpdateActor / StepActor, :271-284 BuildActorProblemY."""

    def __init__(self, critic, critic_mults, actor, actor_mults, S, A, mem_size, batch, actor_batch, discount, num_init_samples, solver, seed,
                 freeze_target_iters=0, init_input_offset_scale=True):
        self.critic, self.actor_net, self.S, self.A = critic, actor, S, A
        self.batch, self.actor_batch, self.discount, self.freeze = batch, actor_batch, discount, freeze_target_iters
        self.W = 1 + 2 * S + A
        self.mem = np.zeros((mem_size, self.W), np.float32); self.flags = np.zeros(mem_size, np.int64)
        self.mem_size, self.head, self.num = mem_size, 0, 0
        self.num_init_samples, self.solver, self.init_os = num_init_samples, solver, init_input_offset_scale
        cat = lambda net, mults, k: np.concatenate([np.full(n, m[k]) for n, m in zip(net.sizes, mults)])
        self.c_lr, self.c_dec, self.a_lr, self.a_dec = cat(critic, critic_mults, 0), cat(critic, critic_mults, 1), cat(actor, actor_mults, 0), cat(actor, actor_mults, 1)
        self.wc = np.zeros(critic.num_params); self.wc_target = self.wc.copy(); self.hc = np.zeros(critic.num_params)
        self.wa = np.zeros(actor.num_params); self.ha = np.zeros(actor.num_params)
        self.in_off, self.in_scale = np.zeros(S), np.ones(S)
        self.c_out_off, self.c_out_scale = np.full(1, -0.5), np.full(1, 2.0)
        self.a_out_off, self.a_out_scale = np.zeros(A), np.ones(A)
        self.rng = np.random.RandomState(seed)
        self.iter = self.actor_iter = 0; self.stage_train = False
        self.off_policy, self.actor_buf, self.actor_td = [], [], []
        self.last_loss = self.last_actor_loss = None

    def add_tuples(self, rows, flags):
        out = []
        for r, f in zip(rows, flags):
            r = np.asarray(r, np.float32)
            if not np.all(np.isfinite(r)):
                out.append(-1); continue
            t = self.head
            self.mem[t] = r; self.flags[t] = int(f)
            self.head = (self.head + 1) % self.mem_size; self.num = min(self.mem_size, self.num + 1)
            if f & 2:
                if t not in self.off_policy: self.off_policy.append(t)
            else:
                RefTrainerBook._remove(self.off_policy, t)
            while t in self.actor_buf:
                i = self.actor_buf.index(t); last, ltd = self.actor_buf.pop(), self.actor_td.pop()
                if i < len(self.actor_buf): self.actor_buf[i] = last; self.actor_td[i] = ltd
            out.append(t)
        return out

    def v(self, w, X):
        X = np.atleast_2d(np.asarray(X, np.float64))
        return (self.critic.forward(w, (X + self.in_off) * self.in_scale) / self.c_out_scale - self.c_out_off)[:, 0]

    def _target(self):
        return self.wc_target if self.freeze > 0 else self.wc

    def new_v(self, ids):
        S, A, g = self.S, self.A, self.discount
        v_end = self.v(self._target(), self.mem[ids, 1 + S + A:].astype(np.float64))
        r = self.mem[ids, 0].astype(np.float64) * (1.0 - g)
        return np.where(self.flags[ids] & 1, r, r + g * v_end)

    def _sgd(self, net, w, h, lr, dec, X, Y, out_off, out_scale):
        x = (np.asarray(X, np.float64) + self.in_off) * self.in_scale
        label = (Y + out_off) * out_scale
        out = net.forward(w, x, keep=True)
        loss = 0.5 * ((out - label) ** 2).sum() / x.shape[0]
        grad = net.backward((out - label) / x.shape[0])
        s = self.solver
        w, h = caffe_sgd_step(w, grad, h, s["base_lr"], s["momentum"], s["weight_decay"], lr, dec)
        return w, h, loss

    def train(self):
        if not self.stage_train and self.num >= self.num_init_samples and self.num > 0:
            if self.num_init_samples > 1 and self.init_os:
                X = self.mem[:self.num, 1:1 + self.S].astype(np.float64)
                mean = X.mean(0); std = np.sqrt(((X - mean) ** 2).mean(0))
                self.in_off, self.in_scale = -mean, np.where(std == 0, 0.0, 1.0 / np.where(std == 0, 1.0, std))
            self.stage_train = True
        if not self.stage_train:
            return
        S, A = self.S, self.A
        ids = [int(self.rng.randint(0, self.num)) for _ in range(self.batch)]
        self.wc, self.hc, self.last_loss = self._sgd(self.critic, self.wc, self.hc, self.c_lr, self.c_dec, self.mem[ids, 1:1 + S], self.new_v(ids)[:, None], self.c_out_off, self.c_out_scale)
        # UpdateActor
        n = len(self.off_policy)
        drawn = []
        for _ in range(min(self.actor_batch, n)):
            t = self.off_policy[int(self.rng.randint(0, n))]
            if t not in self.actor_buf and t not in drawn:
                drawn.append(t)
        if drawn:
            td = self.new_v(drawn) - self.v(self._target(), self.mem[drawn, 1:1 + S].astype(np.float64))
            for t, d in zip(drawn, td):
                if d > 0:
                    self.actor_buf.append(t); self.actor_td.append(float(d))
        for _ in range(len(self.actor_buf) // self.actor_batch):
            b = self.actor_buf[:self.actor_batch]
            self.wa, self.ha, self.last_actor_loss = self._sgd(self.actor_net, self.wa, self.ha, self.a_lr, self.a_dec, self.mem[b, 1:1 + S],
                                                               self.mem[b, 1 + S:1 + S + A].astype(np.float64), self.a_out_off, self.a_out_scale)
            self.actor_iter += 1
            del self.actor_buf[:self.actor_batch]; del self.actor_td[:self.actor_batch]
        if self.freeze > 0 and self.iter > 0 and self.iter % self.freeze == 0:
            self.wc_target = self.wc.copy()
        self.iter += 1
