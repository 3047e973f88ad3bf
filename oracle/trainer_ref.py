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
