"""TEST INFRASTRUCTURE (oracle/): the reference's own TRAINERS, compiled unchanged, driven from Python.

oracle/_ref/libref_learn.so = /root/reference/learning/{NeuralNetTrainer, MACETrainer, QNetTrainer, ACTrainer, CaclaTrainer, NeuralNetLearner, ACLearner, ParamServer,
TrainerInterface, ExpTuple}.cpp built by oracle/_ref_build/Makefile against the reference's own learning/NeuralNet.h. What cannot be compiled is learning/NeuralNet.cpp
(Caffe); its cNeuralNet methods are defined in oracle/_ref_build/ref_learn_net.cpp over the callbacks installed here: the numpy fp64 networks (forward, hand-derived
backward) and the Caffe SGD rule of oracle/trainer_ref.py. So everything the TRAINERS decide -- replay slots, critic / actor buffers with their move-last-into-hole
removal, minibatch draws from cMathUtil::gRand, BuildProblemY labels, the new_q > Q_target(s) filter, UpdateActorBatchBuffer's candidate order, target refreshes, the
stage switch with CalcOffsetScale -- is the reference's own code, and only the inside of a network pass is a restatement.

Only tests/ use this module (tests/test_reference_learn.py); the frozen traces it produces (tests/golden/make_ref_golden_learn.py) travel to the GPU box.
"""
import ctypes as C
import os

import numpy as np

from . import trainer_ref as ref

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libref_learn.so")
# the same reference trainers with the PRODUCT's nets behind cNeuralNet (oracle/_ref_build/ref_learn_net_native.cpp over include/BatchNeuralNet.h): the plain-loop
# check build of the native step on the CPU box, lib/libdtrl.so on the GPU box
NATIVE_LIB_PATH = os.path.join(HERE, "_ref", "libref_learn_native.so")
NATIVE_HIP_LIB_PATH = os.path.join(HERE, "_ref", "libref_learn_native_hip.so")
_libs = {}

NET_NEW = C.CFUNCTYPE(C.c_int, C.c_char_p)
NET_FREE = C.CFUNCTYPE(None, C.c_int)
NET_DIMS = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int))
SOLVER_LOAD = C.CFUNCTYPE(C.c_int, C.c_int, C.c_char_p)
SOLVER_RESET = C.CFUNCTYPE(None, C.c_int)
NET_FORWARD = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double))
NET_STEP = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int)
NET_COPY = C.CFUNCTYPE(None, C.c_int, C.c_int)


class HarnessStruct(C.Structure):
    _fields_ = [("net_new", NET_NEW), ("net_free", NET_FREE), ("net_dims", NET_DIMS), ("solver_load", SOLVER_LOAD), ("solver_reset", SOLVER_RESET),
                ("net_forward", NET_FORWARD), ("net_step", NET_STEP), ("net_copy", NET_COPY)]


class Params(C.Structure):
    _fields_ = [("net_file", C.c_char_p), ("solver_file", C.c_char_p), ("actor_net_file", C.c_char_p), ("actor_solver_file", C.c_char_p),
                ("playback_mem_size", C.c_int), ("pool_size", C.c_int), ("num_init_samples", C.c_int), ("num_steps_per_iter", C.c_int),
                ("freeze_target_iters", C.c_int), ("init_input_offset_scale", C.c_int), ("discount", C.c_double),
                ("num_action_frags", C.c_int), ("action_frag_size", C.c_int)]


def available():
    return os.path.exists(LIB_PATH)


def lib(path=None):
    path = path or LIB_PATH
    if path not in _libs:
        L = C.CDLL(path)
        vp = C.c_void_p
        L.ref_learn_set_harness.argtypes = [C.POINTER(HarnessStruct)]
        L.ref_learn_seed_rand.argtypes = [C.c_ulong]
        L.ref_learn_rand_new.restype = vp; L.ref_learn_rand_new.argtypes = [C.c_ulong]
        L.ref_learn_rand_free.argtypes = [vp]
        L.ref_learn_rand_int.argtypes = [vp, C.c_int, C.c_int]
        L.ref_learn_rand_double.restype = C.c_double; L.ref_learn_rand_double.argtypes = [vp, C.c_double, C.c_double]
        L.ref_learn_trainer_create.restype = vp; L.ref_learn_trainer_create.argtypes = [C.c_int, C.POINTER(Params)]
        L.ref_learn_trainer_destroy.argtypes = [vp]
        L.ref_learn_add_tuple.argtypes = [vp, C.c_double, C.c_uint, vp, vp, vp, C.c_int, C.c_int]
        L.ref_learn_learner_train.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int]
        L.ref_learn_learner_iter.argtypes = [vp]; L.ref_learn_learner_num_tuples.argtypes = [vp]
        for name in ("train",):
            getattr(L, "ref_learn_" + name).argtypes = [vp]
        for name in ("iter", "actor_iter", "stage", "head", "num_stored", "num_tuples", "batch_size", "state_size", "action_size", "num_pool", "actor_net", "mem_cols"):
            getattr(L, "ref_learn_" + name).argtypes = [vp]
        L.ref_learn_pool_net.argtypes = [vp, C.c_int]
        L.ref_learn_buffer.argtypes = [vp, C.c_int, vp, C.c_int]
        L.ref_learn_mem_row.argtypes = [vp, C.c_int, vp, vp]
        for name in ("set_input_offset_scale", "set_output_offset_scale", "set_actor_output_offset_scale", "set_critic_output_offset_scale", "get_input_offset_scale"):
            getattr(L, "ref_learn_" + name).argtypes = [vp, vp, vp, C.c_int]
        if hasattr(L, "ref_learn_native_config"):
            L.ref_learn_native_config.argtypes = [C.c_char_p, C.c_int]
            L.ref_learn_native_num_params.restype = C.c_longlong; L.ref_learn_native_num_params.argtypes = [C.c_int]
            L.ref_learn_native_get_params.argtypes = [C.c_int, vp, C.c_longlong]
            L.ref_learn_native_set_params.argtypes = [C.c_int, vp, C.c_longlong]
        _libs[path] = L
    return _libs[path]


class RefRandStream:
    """An independent cRand (util/Rand.cpp, compiled from the reference) with numpy RandomState's `randint(lo, hi)` call shape: handed to the product's trainer
    (or the numpy restatement) with the seed cMathUtil::gRand got, it makes their index draws the reference's draw for draw."""

    def __init__(self, seed):
        self._L = lib()
        self._h = self._L.ref_learn_rand_new(int(seed))

    def randint(self, lo, hi=None, size=None):
        if hi is None:
            lo, hi = 0, lo
        if size is not None:
            return np.array([self.randint(lo, hi) for _ in range(size)], np.int64)
        return int(self._L.ref_learn_rand_int(self._h, int(lo), int(hi)))

    def __del__(self):
        try:
            self._L.ref_learn_rand_free(self._h)
        except Exception:
            pass


class HarnessNet:
    """One cNeuralNet of the reference as the harness sees it: a numpy net (oracle/trainer_ref.py), ONE parameter vector, the solver's history."""

    def __init__(self, net, mults, solver, batch):
        self.net, self.solver, self.batch = net, solver, batch
        self.lr_mult = np.concatenate([np.full(n, m[0]) for n, m in zip(net.sizes, mults)])
        self.decay_mult = np.concatenate([np.full(n, m[1]) for n, m in zip(net.sizes, mults)])
        self.w = np.zeros(net.num_params); self.hist = np.zeros(net.num_params)
        self.in_size = net.n_terrain + net.n_char
        self.out_size = None
        self.has_solver = False
        self.steps = []      # (x, labels) of every solver step, normalised coordinates: the minibatches and labels the reference built
        self.last_loss = None


class Harness:
    """Installs the callbacks. make_net(net_file: str) -> HarnessNet decides the topology / multipliers / solver constants for a prototxt path."""

    def __init__(self, make_net):
        self.make_net = make_net
        self.nets = {}
        self._next = 0
        self.log_steps = True
        self._cb = HarnessStruct(NET_NEW(self._new), NET_FREE(self._free), NET_DIMS(self._dims), SOLVER_LOAD(self._solver_load), SOLVER_RESET(self._solver_reset),
                                 NET_FORWARD(self._forward), NET_STEP(self._step), NET_COPY(self._copy))
        lib().ref_learn_set_harness(C.byref(self._cb))

    def _new(self, net_file):
        i = self._next; self._next += 1
        self.nets[i] = self.make_net(net_file.decode())
        return i

    def _free(self, i):
        self.nets.pop(i, None)

    def _dims(self, i, pin, pout):
        n = self.nets[i]
        pin[0] = n.in_size; pout[0] = n.out_size

    def _solver_load(self, i, solver_file):
        self.nets[i].has_solver = True
        return self.nets[i].batch

    def _solver_reset(self, i):
        self.nets[i].hist[:] = 0

    def _forward(self, i, x, n, y):
        h = self.nets[i]
        X = np.ctypeslib.as_array(x, shape=(n, h.in_size))
        Y = h.net.forward(h.w, X)
        np.ctypeslib.as_array(y, shape=(n, h.out_size))[:] = Y

    def _step(self, i, x, y, n, iters):
        h = self.nets[i]
        X = np.ctypeslib.as_array(x, shape=(n, h.in_size)).copy()
        L = np.ctypeslib.as_array(y, shape=(n, h.out_size)).copy()
        if self.log_steps:
            h.steps.append((X, L))
        s = h.solver
        for _ in range(iters):
            out = h.net.forward(h.w, X, keep=True)
            h.last_loss = 0.5 * ((out - L) ** 2).sum() / n            # EuclideanLoss: 1 / (2 N) sum ||y - label||^2
            grad = h.net.backward((out - L) / n)
            h.w, h.hist = ref.caffe_sgd_step(h.w, grad, h.hist, s["base_lr"], s["momentum"], s["weight_decay"], h.lr_mult, h.decay_mult)

    def _copy(self, dst, src):
        self.nets[dst].w = self.nets[src].w.copy()


KIND = {"mace": 0, "q": 1, "cacla": 2}


class RefTrainer:
    """cMACETrainer / cQNetTrainer / cCaclaTrainer of the reference behind ref_learn_api.cpp's C ABI."""

    def __init__(self, kind, harness, net_file, solver_file, mem_size, num_init_samples, discount, freeze_target_iters=0, init_input_offset_scale=True,
                 num_frags=1, frag_size=1, actor_net_file="", actor_solver_file="", seed=0, lib_path=None):
        self._L = lib(lib_path)
        self.harness = harness
        self._L.ref_learn_seed_rand(int(seed))
        p = Params(net_file.encode(), solver_file.encode(), actor_net_file.encode(), actor_solver_file.encode(), int(mem_size), 1, int(num_init_samples), 1,
                   int(freeze_target_iters), int(bool(init_input_offset_scale)), float(discount), int(num_frags), int(frag_size))
        self._keep = p
        self._h = self._L.ref_learn_trainer_create(KIND[kind], C.byref(p))
        self.S = self._L.ref_learn_state_size(self._h)
        self.A = self._L.ref_learn_action_size(self._h)
        self.W = self._L.ref_learn_mem_cols(self._h)

    def close(self):
        if self._h:
            self._L.ref_learn_trainer_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_rows(self, rows, flags):
        """rows in the MACE replay layout [r | s | a | s'] (what the rollout engine's tuple drain emits), flags = tExpTuple::mFlags. Returns the slots."""
        S, A = self.S, self.A
        out = []
        for r, f in zip(np.asarray(rows, np.float64), flags):
            sb = np.ascontiguousarray(r[1:1 + S]); ac = np.ascontiguousarray(r[1 + S:1 + S + A]); se = np.ascontiguousarray(r[1 + S + A:1 + 2 * S + A])
            out.append(self._L.ref_learn_add_tuple(self._h, float(r[0]), int(f), sb.ctypes.data, se.ctypes.data, ac.ctypes.data, S, A))
        return out

    def train(self): self._L.ref_learn_train(self._h)

    def learner_train(self, rows, flags):
        """cNeuralNetLearner::Train(tuples) of the reference (learning/NeuralNetLearner.cpp:33-46): AddTuples + Train + SyncNet through a learner the trainer handed out
        (RequestLearner), exactly what cScenarioExp calls when its tuple buffer is full. Returns the harness net the learner synchronised (the env side's policy)."""
        S, A = self.S, self.A
        r = np.ascontiguousarray(np.asarray(rows, np.float64))
        rew = np.ascontiguousarray(r[:, 0]); sb = np.ascontiguousarray(r[:, 1:1 + S]); ac = np.ascontiguousarray(r[:, 1 + S:1 + S + A]); se = np.ascontiguousarray(r[:, 1 + S + A:1 + 2 * S + A])
        fl = np.ascontiguousarray(np.asarray(flags, np.uint32))
        nid = self._L.ref_learn_learner_train(self._h, len(r), rew.ctypes.data, fl.ctypes.data, sb.ctypes.data, se.ctypes.data, ac.ctypes.data, S, A)
        return self.harness.nets[nid] if self.harness is not None and hasattr(self.harness, "nets") else nid

    @property
    def learner_iter(self): return self._L.ref_learn_learner_iter(self._h)
    @property
    def learner_num_tuples(self): return self._L.ref_learn_learner_num_tuples(self._h)
    @property
    def iter(self): return self._L.ref_learn_iter(self._h)
    @property
    def actor_iter(self): return self._L.ref_learn_actor_iter(self._h)
    @property
    def stage_train(self): return self._L.ref_learn_stage(self._h) == 1
    @property
    def head(self): return self._L.ref_learn_head(self._h)
    @property
    def num_stored(self): return self._L.ref_learn_num_stored(self._h)
    @property
    def batch(self): return self._L.ref_learn_batch_size(self._h)

    def pool_net(self, i=0):
        return self.harness.nets[self._L.ref_learn_pool_net(self._h, i)]

    # native variant (libref_learn_native*.so): the pool nets are the product's; their Caffe-blob-order weight vectors by pool index
    def pool_weights(self, i=0):
        nid = self._L.ref_learn_pool_net(self._h, i)
        w = np.zeros(int(self._L.ref_learn_native_num_params(nid)), np.float32)
        assert self._L.ref_learn_native_get_params(nid, w.ctypes.data, w.size) == 0
        return w

    def set_pool_weights(self, i, w):
        w = np.ascontiguousarray(w, np.float32)
        assert self._L.ref_learn_native_set_params(self._L.ref_learn_pool_net(self._h, i), w.ctypes.data, w.size) == 0

    def num_pool(self): return self._L.ref_learn_num_pool(self._h)

    def actor_net(self):
        return self.harness.nets[self._L.ref_learn_actor_net(self._h)]

    def buffer(self, which):
        """0 = critic buffer, 1 = actor (exploration / off-policy) buffer, 2 = actor batch buffer; None if the trainer has none"""
        cap = 1 << 16
        a = np.zeros(cap, np.int32)
        n = self._L.ref_learn_buffer(self._h, which, a.ctypes.data, cap)
        return None if n < 0 else a[:n].tolist()

    def mem_row(self, t):
        a = np.zeros(self.W, np.float32); f = C.c_uint()
        self._L.ref_learn_mem_row(self._h, int(t), a.ctypes.data, C.byref(f))
        return a, f.value

    def _osc(self, fn, off, scale):
        off = np.ascontiguousarray(off, np.float64); scale = np.ascontiguousarray(scale, np.float64)
        getattr(self._L, "ref_learn_" + fn)(self._h, off.ctypes.data, scale.ctypes.data, off.size)

    def set_input_offset_scale(self, off, scale): self._osc("set_input_offset_scale", off, scale)
    def set_output_offset_scale(self, off, scale): self._osc("set_output_offset_scale", off, scale)
    def set_actor_output_offset_scale(self, off, scale): self._osc("set_actor_output_offset_scale", off, scale)
    def set_critic_output_offset_scale(self, off, scale): self._osc("set_critic_output_offset_scale", off, scale)

    def input_offset_scale(self):
        off = np.zeros(self.S); sc = np.zeros(self.S)
        self._L.ref_learn_get_input_offset_scale(self._h, off.ctypes.data, sc.ctypes.data, self.S)
        return off, sc
