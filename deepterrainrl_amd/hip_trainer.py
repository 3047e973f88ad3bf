"""MI355X-native MACE trainer step: cMACETrainer's iteration on hand-written HIP kernels (include/dtrl_trainer.h, csrc/dtrl_trainer.hip).

`HipMACETrainer` keeps cMACETrainer's host-side bookkeeping exactly as `trainer.MACETrainer` has it (replay slots, critic / actor index buffers with
the reference's move-last-into-hole removal, minibatch draws, stages, target refresh schedule) and replaces everything that touched the network:

  Step()       critic: slots -> page-locked index array -> dtrl_trainer_critic_step: gather + normalise s and s' from the replay rows, Q_target(s'),
               forward of the current net, labels (new_q over the taken fragment's value), EuclideanLoss gradient, backward, Caffe SGD update
  UpdateActor  dtrl_trainer_actor_filter (new_q > Q_target(s) for the candidates, mask read from page-locked memory) and dtrl_trainer_actor_step
  Eval / _solver_step   dtrl_trainer_eval / dtrl_trainer_step for callers that bring their own batches

One iteration is ~65 kernel launches on one stream and ONE host wait (the actor mask); weights, history, activations and the replay rows stay on the
device, no framework op runs inside an iteration. `trainer.MACETrainer` (PyTorch ops replayed as HIP graphs) stays as the peer the tests compare
against, the whole-trainer numpy restatement (test infrastructure) as the oracle. The product path is lib/libdtrl.so; there is no CPU fallback (tests bind the plain-loop check build
of the same operand definitions from tests/emul through `lib_path`).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import LIB_PATH, DtrlError
from .trainer import MACETrainer, QNetTrainer, CaclaTrainer

TRAINER_ABI_SYMBOLS = [
    "dtrl_trainer_create", "dtrl_trainer_destroy", "dtrl_trainer_last_error", "dtrl_trainer_set_stream", "dtrl_trainer_sync", "dtrl_trainer_num_params",
    "dtrl_trainer_set_params", "dtrl_trainer_get_params", "dtrl_trainer_params_device", "dtrl_trainer_set_normalizers", "dtrl_trainer_update_target",
    "dtrl_trainer_eval", "dtrl_trainer_step", "dtrl_trainer_bind_replay", "dtrl_trainer_idx", "dtrl_trainer_better", "dtrl_trainer_loss",
    "dtrl_trainer_critic_step", "dtrl_trainer_actor_filter", "dtrl_trainer_actor_step", "dtrl_trainer_debug_get", "dtrl_trainer_critic_step_and_filter",
    "dtrl_trainer_stage_rows", "dtrl_trainer_stage_flags", "dtrl_trainer_stage_capacity", "dtrl_trainer_add_staged",
    "dtrl_trainer_create_from_files", "dtrl_trainer_dims", "dtrl_trainer_init_xavier", "dtrl_trainer_eval_host", "dtrl_trainer_step_host", "dtrl_trainer_get_normalizers",
    "dtrl_trainer_copy_model", "dtrl_trainer_bind_grad", "dtrl_trainer_grad_device", "dtrl_trainer_grad_step", "dtrl_trainer_critic_grad", "dtrl_trainer_actor_grad", "dtrl_trainer_zero_grad", "dtrl_trainer_apply_grad",
    "dtrl_trainer_value_step", "dtrl_trainer_td_filter", "dtrl_trainer_td", "dtrl_trainer_action_step",
]


class TrainerDesc(C.Structure):
    _fields_ = [("state_size", C.c_int32), ("n_terrain", C.c_int32), ("conv_ch", C.c_int32 * 3), ("conv_k", C.c_int32 * 3),
                ("fc_terr", C.c_int32), ("fc_trunk", C.c_int32), ("fc_head", C.c_int32), ("n_heads", C.c_int32), ("head_out", C.c_int32 * 8),
                ("n_frags", C.c_int32), ("frag_size", C.c_int32), ("batch", C.c_int32), ("max_eval", C.c_int32),
                ("base_lr", C.c_float), ("momentum", C.c_float), ("weight_decay", C.c_float), ("discount", C.c_float), ("freeze_target", C.c_int32)]


def _bind(path):
    if not os.path.exists(path):
        raise DtrlError("HIP extension missing: %s (run __graft_entry__.build() / make -C deepterrainrl_amd/csrc)" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.dtrl_trainer_create.argtypes = [C.POINTER(TrainerDesc), C.c_int, C.POINTER(vp)]
    L.dtrl_trainer_destroy.argtypes = [vp]
    L.dtrl_trainer_last_error.restype = C.c_char_p; L.dtrl_trainer_last_error.argtypes = [vp]
    L.dtrl_trainer_set_stream.argtypes = [vp, vp]
    L.dtrl_trainer_sync.argtypes = [vp]
    L.dtrl_trainer_num_params.restype = C.c_int64; L.dtrl_trainer_num_params.argtypes = [vp]
    L.dtrl_trainer_set_params.argtypes = [vp, C.c_int, vp, C.c_int64]
    L.dtrl_trainer_get_params.argtypes = [vp, C.c_int, vp, C.c_int64]
    L.dtrl_trainer_params_device.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.dtrl_trainer_debug_get.argtypes = [vp, C.c_int, vp, C.c_int64]
    L.dtrl_trainer_set_normalizers.argtypes = [vp, vp, vp, vp, vp]
    L.dtrl_trainer_update_target.argtypes = [vp]
    L.dtrl_trainer_eval.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.dtrl_trainer_step.argtypes = [vp, vp, vp]
    L.dtrl_trainer_bind_replay.argtypes = [vp, vp, vp, C.c_int]
    for name, ty in (("dtrl_trainer_idx", C.c_int64), ("dtrl_trainer_better", C.c_int32), ("dtrl_trainer_loss", C.c_float)):
        getattr(L, name).restype = C.POINTER(ty); getattr(L, name).argtypes = [vp]
    L.dtrl_trainer_stage_rows.restype = C.POINTER(C.c_float); L.dtrl_trainer_stage_rows.argtypes = [vp]
    L.dtrl_trainer_stage_flags.restype = C.POINTER(C.c_int64); L.dtrl_trainer_stage_flags.argtypes = [vp]
    L.dtrl_trainer_stage_capacity.argtypes = [vp]
    L.dtrl_trainer_add_staged.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.c_int64]
    L.dtrl_trainer_critic_step.argtypes = [vp]
    L.dtrl_trainer_critic_step_and_filter.argtypes = [vp]
    L.dtrl_trainer_actor_filter.argtypes = [vp, C.c_int]
    L.dtrl_trainer_actor_step.argtypes = [vp]
    L.dtrl_trainer_bind_grad.argtypes = [vp, vp]
    L.dtrl_trainer_grad_device.argtypes = [vp, C.POINTER(vp)]
    L.dtrl_trainer_grad_step.argtypes = [vp, vp, vp]
    for name in ("critic_grad", "actor_grad", "zero_grad"):
        getattr(L, "dtrl_trainer_" + name).argtypes = [vp]
    L.dtrl_trainer_apply_grad.argtypes = [vp, C.c_int]
    L.dtrl_trainer_value_step.argtypes = [vp, C.c_int]
    L.dtrl_trainer_td_filter.argtypes = [vp, C.c_int]
    L.dtrl_trainer_td.restype = C.POINTER(C.c_float); L.dtrl_trainer_td.argtypes = [vp]
    L.dtrl_trainer_action_step.argtypes = [vp]
    return L


def desc_from_net(desc, state_size, batch, max_eval, solver, discount, freeze_target):
    """dtrl_trainer_desc from trainer.parse_net / parse_solver output."""
    d = TrainerDesc()
    d.state_size, d.n_terrain = state_size, desc["n_terrain"]
    for l, cv in enumerate(desc["convs"]):
        d.conv_ch[l], d.conv_k[l] = cv["num_output"], cv["kernel_w"]
    ips = desc["ips"]
    d.fc_terr = ips["terr_ip0"]["num_output"]
    if desc["n_frags"] > 0:
        d.fc_trunk, d.fc_head = ips["ip0"]["num_output"], ips["val_ip0"]["num_output"]
        d.n_heads = 1 + desc["n_frags"]
        d.head_out[0] = desc["n_frags"]
        for f in range(desc["n_frags"]):
            d.head_out[1 + f] = desc["frag_size"]
    else:
        d.fc_trunk, d.fc_head, d.n_heads = ips["ip1"]["num_output"], ips["ip2"]["num_output"], 1
        d.head_out[0] = desc["frag_size"]
    d.n_frags, d.frag_size = desc["n_frags"], desc["frag_size"]
    d.batch, d.max_eval = batch, max_eval
    d.base_lr, d.momentum, d.weight_decay, d.discount = solver["base_lr"], solver["momentum"], solver["weight_decay"], discount
    d.freeze_target = 1 if freeze_target else 0
    return d


class NativeTrainer:
    """Thin handle over the C ABI (one dtrl_trainer)."""

    def __init__(self, cdesc, device_id=-1, lib_path=None):
        self._lib = _bind(lib_path or LIB_PATH)
        h = C.c_void_p()
        rc = self._lib.dtrl_trainer_create(C.byref(cdesc), int(device_id), C.byref(h))
        if rc != 0:
            raise DtrlError("dtrl_trainer_create failed (%d): %s" % (rc, self._lib.dtrl_trainer_last_error(None).decode()))
        self._h = h
        self.num_params = int(self._lib.dtrl_trainer_num_params(h))
        self.batch, self.max_eval = int(cdesc.batch), int(cdesc.max_eval)
        self.idx = np.ctypeslib.as_array(self._lib.dtrl_trainer_idx(h), shape=(2 * self.max_eval,))
        self.better = np.ctypeslib.as_array(self._lib.dtrl_trainer_better(h), shape=(self.max_eval,))
        self.loss = np.ctypeslib.as_array(self._lib.dtrl_trainer_loss(h), shape=(4,))
        self.td = np.ctypeslib.as_array(self._lib.dtrl_trainer_td(h), shape=(self.max_eval,))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dtrl_trainer_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise DtrlError("dtrl_trainer call failed (%d): %s" % (rc, self._lib.dtrl_trainer_last_error(self._h).decode()))

    def set_stream(self, ptr): self._chk(self._lib.dtrl_trainer_set_stream(self._h, C.c_void_p(ptr)))
    def sync(self): self._chk(self._lib.dtrl_trainer_sync(self._h))

    def set_params(self, which, a):
        a = np.ascontiguousarray(a, np.float32); self._chk(self._lib.dtrl_trainer_set_params(self._h, which, a.ctypes.data_as(C.c_void_p), a.size))

    def get_params(self, which):
        a = np.zeros(self.num_params, np.float32); self._chk(self._lib.dtrl_trainer_get_params(self._h, which, a.ctypes.data_as(C.c_void_p), a.size)); return a

    def debug_get(self, which, n):
        a = np.zeros(int(n), np.float32); self._chk(self._lib.dtrl_trainer_debug_get(self._h, which, a.ctypes.data_as(C.c_void_p), a.size)); return a

    def params_device(self, which=0):
        p = C.c_void_p(); self._chk(self._lib.dtrl_trainer_params_device(self._h, which, C.byref(p))); return p.value

    def set_normalizers(self, io=None, isc=None, oo=None, osc=None):
        arrs = [None if a is None else np.ascontiguousarray(a, np.float64) for a in (io, isc, oo, osc)]
        self._chk(self._lib.dtrl_trainer_set_normalizers(self._h, *[None if a is None else a.ctypes.data_as(C.c_void_p) for a in arrs]))

    def update_target(self): self._chk(self._lib.dtrl_trainer_update_target(self._h))
    def eval(self, which, x_ptr, n, y_ptr): self._chk(self._lib.dtrl_trainer_eval(self._h, which, C.c_void_p(x_ptr), n, C.c_void_p(y_ptr)))
    def step(self, x_ptr, y_ptr): self._chk(self._lib.dtrl_trainer_step(self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr)))
    def bind_replay(self, mem_ptr, flags_ptr, W):
        self._chk(self._lib.dtrl_trainer_bind_replay(self._h, C.c_void_p(mem_ptr), C.c_void_p(flags_ptr), W))
        self.stage_cap = self._lib.dtrl_trainer_stage_capacity(self._h)
        pr, pf = self._lib.dtrl_trainer_stage_rows(self._h), self._lib.dtrl_trainer_stage_flags(self._h)
        if not pr or not pf:
            raise DtrlError("trainer staging area: " + (self._lib.dtrl_trainer_last_error(self._h) or b"allocation failed").decode())
        self.stage_rows = np.ctypeslib.as_array(pr, shape=(self.stage_cap, W))
        self.stage_flags = np.ctypeslib.as_array(pf, shape=(self.stage_cap,))

    def add_staged(self, first, n, head, mem_size): self._chk(self._lib.dtrl_trainer_add_staged(self._h, first, n, head, mem_size))
    def critic_step(self): self._chk(self._lib.dtrl_trainer_critic_step(self._h))
    def critic_step_and_filter(self): self._chk(self._lib.dtrl_trainer_critic_step_and_filter(self._h))
    def actor_filter(self, n): self._chk(self._lib.dtrl_trainer_actor_filter(self._h, n))
    def actor_step(self): self._chk(self._lib.dtrl_trainer_actor_step(self._h))
    # data-parallel step (include/dtrl_trainer.h): gradient only / update from the all-reduced gradient
    def bind_grad(self, ptr): self._chk(self._lib.dtrl_trainer_bind_grad(self._h, C.c_void_p(ptr) if ptr else None))
    def grad_device(self):
        p = C.c_void_p(); self._chk(self._lib.dtrl_trainer_grad_device(self._h, C.byref(p))); return p.value
    def grad_step(self, x_ptr, y_ptr): self._chk(self._lib.dtrl_trainer_grad_step(self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr)))
    def critic_grad(self): self._chk(self._lib.dtrl_trainer_critic_grad(self._h))
    def actor_grad(self): self._chk(self._lib.dtrl_trainer_actor_grad(self._h))
    def zero_grad(self): self._chk(self._lib.dtrl_trainer_zero_grad(self._h))
    def apply_grad(self, slot): self._chk(self._lib.dtrl_trainer_apply_grad(self._h, int(slot)))
    def value_step(self, kind): self._chk(self._lib.dtrl_trainer_value_step(self._h, int(kind)))
    def td_filter(self, n): self._chk(self._lib.dtrl_trainer_td_filter(self._h, int(n)))
    def action_step(self): self._chk(self._lib.dtrl_trainer_action_step(self._h))


class _HipNetSide:
    """The network side of a trainer.py trainer on the native step (mixin in front of MACETrainer / QNetTrainer / CaclaTrainer): weights, solver history and
    normalisers live in a dtrl_trainer, _eval / _solver_step are dtrl_trainer_eval / dtrl_trainer_step, new tuples are stored from the page-locked staging
    area. float32 only (the shipped nets are float32-valued)."""

    def __init__(self, *a, lib_path=None, **kw):
        kw.setdefault("use_graphs", False)      # the torch peer's graph machinery is not used: nothing framework-side runs inside an iteration
        kw["dtype"] = torch.float32
        self._trainer_lib = lib_path               # (CaclaTrainer.Reset, called by the base constructor, builds the actor's trainer with it)
        super().__init__(*a, **kw)
        if self.solver["lr_policy"] != "fixed":
            raise DtrlError("the native trainer step implements lr_policy \"fixed\" (what the shipped solver prototxts use)")
        cdesc = desc_from_net(self.desc, self.S, self.batch, 3 * self.batch, self.solver, self.discount, self.freeze_target_iters > 0)
        dev_id = self.device.index if self.device.type == "cuda" and self.device.index is not None else -1
        self.nt = NativeTrainer(cdesc, dev_id, lib_path)
        assert self.nt.num_params == self.net.num_params()
        # a stream of the trainer's own (its fixed launch sequences are recorded as HIP graphs, which the legacy default stream does not allow); ordered
        # against the framework's stream where the two meet: replay rows written by AddTuples, tensors handed to / taken from _eval and _solver_step
        self._stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None   # (a high-priority stream was measured: no gain beside the rollout)
        if self._stream is not None:
            self.nt.set_stream(self._stream.cuda_stream)
        self.nt.set_params(0, self.net.get_flat())
        self.nt.set_params(3, self.rate_mult.detach().cpu().numpy()); self.nt.set_params(4, self.decay_mult.detach().cpu().numpy())
        self.nt.update_target()
        self.nt.bind_replay(self.mem.data_ptr(), self.flags_dev.data_ptr(), self.W)
        self._push_norm()

    def UseStream(self, stream_ptr):
        """Run the trainer's launches on a stream made elsewhere (a hipStream_t as an int) -- the rollout engine's calibrated side stream
        (BatchScenario.SideStream): its kernels then start on the compute units the frame launches leave free instead of queueing behind them."""
        if self.device.type != "cuda":
            return
        self.nt.sync()
        self._stream = torch.cuda.ExternalStream(int(stream_ptr), device=self.device)
        self.nt.set_stream(int(stream_ptr))

    # ---- state that lives in the native trainer ----
    def _push_norm(self):
        f = lambda t: t.detach().to(torch.float64).cpu().numpy()
        self.nt.set_normalizers(f(self.in_off), f(self.in_scale), f(self.out_off), f(self.out_scale))

    def SetInputOffsetScale(self, off, scale):
        super().SetInputOffsetScale(off, scale); self._push_norm()

    def SetOutputOffsetScale(self, off, scale):
        super().SetOutputOffsetScale(off, scale); self._push_norm()

    def GetWeights(self):
        return self.nt.get_params(0)

    def SetWeights(self, w):
        self.nt.set_params(0, np.asarray(w, np.float32)); self.nt.update_target()

    def WeightsDevicePtr(self):
        """device pointer of the current net's flat weights (Caffe blob order): dtrl_set_policy_device takes it as it is"""
        return self.nt.params_device(0)

    def StreamPtr(self):
        """the trainer's stream as a hipStream_t (int), or None on the CPU check build: dtrl_set_policy_device_on queues the hand-over behind the trainer's work"""
        return int(self._stream.cuda_stream) if self._stream is not None else None

    def UpdateTargetNet(self):
        if hasattr(self, "nt"):
            self.nt.update_target()
        else:
            super().UpdateTargetNet()          # (called once by the base constructor before the native trainer exists)

    def _after_torch(self):
        """the trainer's stream waits for what the framework's current stream has queued (replay writes, input tensors)"""
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream(self.device))

    def _before_torch(self):
        if self._stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._stream)

    def StageTuples(self, rows, flags):
        """A frame's rows into the trainer's page-locked staging area (one host copy); AddTuples(..., staged=k) then stores rows [k, k + n) of it from the
        device side on the trainer's stream. Returns how many rows were staged (the area's capacity bounds it; the caller stages the rest afterwards)."""
        n = min(len(rows), self.nt.stage_cap)
        self.nt.sync()                               # the previous frame's stores have executed
        self.nt.stage_rows[:n] = rows[:n]; self.nt.stage_flags[:n] = flags[:n]
        return n

    def AddTuples(self, rows, flags, staged=None):
        self._staged = staged
        out = super().AddTuples(rows, flags)
        self._staged = None
        return out

    def _store_rows(self, slots, rows, flags, contiguous):
        if self._staged is not None and contiguous and len(slots) <= self.mem_size:
            self.nt.add_staged(self._staged, len(slots), int(slots[0]), self.mem_size)     # (every row of the chunk passed CheckTuple: staged row i -> slot head + i)
            self._staged_dirty = True           # rows written on the TRAINER's stream: a framework read of self.mem / self.flags_dev must be ordered behind them
        else:
            super()._store_rows(slots, rows, flags, contiguous)
            self._replay_dirty = True

    def _order_staged(self):
        """Replay rows stored by add_staged are written asynchronously on the trainer's stream; every framework-side read of self.mem / self.flags_dev (the
        normaliser statistics, HipCaclaTrainer's and HipQNetTrainer's minibatch rows) goes through here first, so that torch's current stream waits for
        them. Beside a running frame those stores can sit in the queue for milliseconds -- without the wait a minibatch could be built from rows not yet
        written. Lazy: one stream wait per batch of stores, none when nothing was staged."""
        if getattr(self, "_staged_dirty", False):
            self._before_torch(); self._staged_dirty = False

    def _idx(self, ids):
        self._order_staged()                    # every framework read of the replay memory indexes it through _idx
        return super()._idx(ids)

    def UpdateOffsetScale(self):
        self._order_staged()                    # cNeuralNet::CalcOffsetScale reads the begin states of every stored tuple
        super().UpdateOffsetScale()

    # ---- network calls ----
    def _eval(self, net, X):
        n = X.shape[0]
        which = 0 if (net is self.net or not self.EnableTargetNet()) else 1
        X = X.to(torch.float32).contiguous()
        self._after_torch()
        out = []
        for k in range(0, n, self.nt.max_eval):
            xs = X[k:k + self.nt.max_eval]
            y = torch.empty((xs.shape[0], self.out_size), dtype=torch.float32, device=self.device)
            self.nt.eval(which, xs.data_ptr(), xs.shape[0], y.data_ptr())
            out.append(y)
        self._before_torch()
        self._keep = (X, out)                  # the kernels read X after this returns
        return out[0] if len(out) == 1 else torch.cat(out)

    def _solver_step(self, X, Y):
        X = X.to(torch.float32).contiguous(); Y = Y.to(torch.float32).contiguous()
        self._after_torch()
        self.nt.step(X.data_ptr(), Y.data_ptr())
        self._keep = (X, Y)
        self.solver_iter += 1
        self.nt.sync()
        return torch.tensor(float(self.nt.loss[0]))


class HipMACETrainer(_HipNetSide, MACETrainer):
    """cMACETrainer with the network side on the native HIP step (see the module docstring). Same constructor as trainer.MACETrainer (+ lib_path)."""

    def OutputModel(self, model_file):
        self.net.set_flat(self.GetWeights())    # the torch net is only a container here
        super().OutputModel(model_file)

    # ---- cMACETrainer::Step on the fused native calls ----
    def Step(self):
        if getattr(self, "_replay_dirty", False):
            self._after_torch(); self._replay_dirty = False
        ids = self.FetchMinibatch(self.batch)
        succ = len(ids) >= self.batch
        if succ and self.EnableTargetNet() and self.stage_train:
            # frozen target: the candidates' test does not depend on the critic update -> one fused pass (dtrl_trainer_critic_step_and_filter). The
            # candidates are drawn here, right behind the critic batch: the same position in the index stream as cMACETrainer::Step's order.
            cand = self.FetchActorMinibatch(self.batch)
            if getattr(self, "_loss_pending", False):
                self.nt.sync(); self._last_loss = float(self.nt.loss[0]); self._loss_pending = False
            B = self.batch
            self.nt.idx[:B] = ids
            self.nt.idx[B:2 * B] = (cand + [ids[0]] * (B - len(cand))) if cand else [ids[0]] * B
            self.nt.critic_step_and_filter()
            self.solver_iter += 1
            self.nt.sync()                      # the one host wait of an iteration
            self._last_loss = float(self.nt.loss[0])
            if cand:
                better = self.nt.better[:len(cand)].copy()
                self.actor_batch_buffer += [t for t, b in zip(cand, better) if b]
            self._run_actor_batches()
            if self.iter > 0 and self.iter % self.freeze_target_iters == 0:
                self.nt.update_target()
            return succ
        if succ:
            if getattr(self, "_loss_pending", False):   # a critic step may still be queued (no actor filter waited since): it must have read its indices
                self.nt.sync(); self._last_loss = float(self.nt.loss[0]); self._loss_pending = False
            self.nt.idx[:self.batch] = ids
            self.nt.critic_step()
            self.solver_iter += 1
            self._loss_pending = True
        self.UpdateActor()
        if self.EnableTargetNet() and self.iter > 0 and self.iter % self.freeze_target_iters == 0:
            self.nt.update_target()
        return succ

    @property
    def last_loss(self):
        if getattr(self, "_loss_pending", False):
            self.nt.sync(); self._last_loss = float(self.nt.loss[0]); self._loss_pending = False
        return None if self._last_loss is None else float(self._last_loss)

    @property
    def last_actor_loss(self):
        if getattr(self, "_aloss_pending", False):
            self.nt.sync(); self._last_actor_loss = float(self.nt.loss[1]); self._aloss_pending = False
        return None if self._last_actor_loss is None else float(self._last_actor_loss)

    def UpdateActorBatchBuffer(self):
        ids = self.FetchActorMinibatch(self.batch)
        if not ids:
            return
        n = len(ids)
        self.nt.idx[self.batch:self.batch + n] = ids   # (the previous filter was waited for, so its window is free)
        self.nt.actor_filter(n)
        self.nt.sync()                          # the one host wait of an iteration: the mask decides what enters the actor batch buffer
        if getattr(self, "_loss_pending", False):
            self._last_loss = float(self.nt.loss[0]); self._loss_pending = False
        better = self.nt.better[:n].copy()
        self.actor_batch_buffer += [t for t, b in zip(ids, better) if b]

    def UpdateActor(self):
        if self.stage_train:
            self.UpdateActorBatchBuffer()
        self._run_actor_batches()

    def _run_actor_batches(self):
        for _ in range(len(self.actor_batch_buffer) // self.batch):
            ids = self.actor_batch_buffer[:self.batch]
            if getattr(self, "_aloss_pending", False):
                self.nt.sync(); self._last_actor_loss = float(self.nt.loss[1]); self._aloss_pending = False
            self.nt.idx[self.nt.max_eval:self.nt.max_eval + self.batch] = ids
            self.nt.actor_step()
            self._aloss_pending = True
            self.solver_iter += 1
            self.actor_iter += 1
            del self.actor_batch_buffer[:self.batch]


class HipMACETrainerDP(HipMACETrainer):
    """cMACETrainer stepped DATA-PARALLEL across the ranks of a process group (last row; the in-scope form of BASELINE configs[3] / [4]): every rank keeps
    the tuples of ITS OWN env shard in its own replay memory and runs the native forward / backward on minibatches drawn from it; per Train() the ranks all-reduce
    (sum) the flat gradient twice -- once for the critic round, once for the actor round -- on the trainer's stream (RCCL; 570 474 + 1 floats = 2.28 MB) and apply
    the identical Caffe SGD update, so the weights stay equal on all ranks with no tuple gather and no weight broadcast. What the reference does for trainer fan-in is a
    pool of learners pushing gradients to a parameter server asynchronously (learning/AsyncMACETrainer.cpp:14-45, learning/ParamServer.cpp:65-90); this is its
    synchronous counterpart.
      * a rank whose critic buffer cannot fill a batch yet (or that holds no full actor batch) contributes a zero gradient with sample count 0: the update divides by
        the total count, so the step is the mean over the samples that exist; a round in which NO rank has a batch changes nothing;
      * ONE actor round per Train() (the reference loops while full batches remain, learning/MACETrainer.cpp:611-626; here a second full batch waits for the next call):
        every rank must issue the same sequence of collectives;
      * the stage switch happens when EVERY rank has its trainer_num_init_samples (all-reduce MIN of a flag), and the input normaliser of the switch is computed from
        the pooled sums of all ranks' begin states (all-reduce of [n, sum x, sum x^2] in float64): cNeuralNet::CalcOffsetScale over the union of the replay memories;
      * iteration counters, target refreshes and the exploration anneal follow the agreed counts and therefore coincide on all ranks.
    `dist`: torch.distributed (initialised); with a one-rank group the trainer equals HipMACETrainer's unfused path step for step (tested)."""

    def __init__(self, *a, dist=None, **kw):
        super().__init__(*a, **kw)
        self.dist = dist
        P = self.nt.num_params
        if self.device.type == "cuda":
            self.grad = torch.zeros(P + 1, dtype=torch.float32, device=self.device)
            self.nt.bind_grad(self.grad.data_ptr())
        else:
            g = (C.c_float * (P + 1)).from_address(self.nt.grad_device())
            self.grad = torch.from_numpy(np.ctypeslib.as_array(g))
        self.world = dist.get_world_size() if dist is not None else 1
        self.collective_s = 0.0
        # DTRL_FORCE_COLLECTIVES=1 (as for sharding.ShardedRollout): issue the gradient all-reduces on a one-rank group too -- the RCCL path on a 1-GPU box
        self.force_collectives = os.environ.get("DTRL_FORCE_COLLECTIVES", "0") == "1"

    def _allreduce_grad(self):
        if self.dist is None or self.world == 1 and not getattr(self, "force_collectives", False):
            return
        from .sharding import all_reduce
        if self._stream is not None:
            with torch.cuda.stream(self._stream):     # behind the gradient kernels on the trainer's stream; the update queued next follows the collective
                all_reduce(self.dist, self.grad)
        else:
            all_reduce(self.dist, self.grad)

    def _agree(self, value, op):
        if self.dist is None or self.world == 1:
            return value
        t = torch.tensor([value], dtype=torch.int64, device=self.device if self.device.type == "cuda" else "cpu")
        from .sharding import all_reduce
        all_reduce(self.dist, t, op=op)
        return int(t.item())

    def UpdateOffsetScale(self):
        if self.dist is None or self.world == 1:
            return super().UpdateOffsetScale()
        self._order_staged()
        X = self.mem[:self.num_tuples, 1:1 + self.S].to(torch.float64)
        pooled = torch.cat([torch.tensor([float(X.shape[0])], dtype=torch.float64, device=X.device), X.sum(0), (X * X).sum(0)])
        if self.dist is not None and self.world > 1:
            from .sharding import all_reduce
            all_reduce(self.dist, pooled)
        n = pooled[0]; mean = pooled[1:1 + self.S] / n
        var = (pooled[1 + self.S:] / n - mean * mean).clamp_min(0.0)
        std = var.sqrt()
        self.SetInputOffsetScale(-mean, torch.where(std == 0, torch.zeros_like(std), 1.0 / std))

    def Train(self):
        if not self.stage_train:
            ready = int(self.num_tuples >= self.num_init_samples and self.num_tuples > 0)
            if self.dist is not None and self.world > 1:
                ready = self._agree(ready, self.dist.ReduceOp.MIN)       # the switch (and its pooled normaliser) happens on all ranks in the same call
            if ready:
                if self.num_init_samples > 1 and self.init_input_offset_scale:
                    self.UpdateOffsetScale()
                self.stage_train = True
        if self.stage_train:
            succ = False
            for _ in range(self.steps_per_iter):
                succ = self.Step()
            if succ:
                self.iter += 1

    def Step(self):
        if getattr(self, "_replay_dirty", False):
            self._after_torch(); self._replay_dirty = False
        B = self.batch
        ids = self.FetchMinibatch(B)
        if len(ids) >= B:
            self.nt.idx[:B] = ids
            self.nt.critic_grad()
        else:
            self.nt.zero_grad()
        self._allreduce_grad()
        self.nt.apply_grad(2)
        # the candidates' test runs against the target net (frozen: unchanged by the update above; not frozen: the updated net, as cMACETrainer::Step orders it)
        self.UpdateActorBatchBuffer()                      # one host wait: the mask (and with it the critic round's sample count)
        n_critic = float(self.nt.loss[2])
        if len(ids) >= B:
            self._last_loss = float(self.nt.loss[0])
        if n_critic > 0:
            self.solver_iter += 1
        if len(self.actor_batch_buffer) >= B:
            self.nt.idx[self.nt.max_eval:self.nt.max_eval + B] = self.actor_batch_buffer[:B]
            self.nt.actor_grad()
            del self.actor_batch_buffer[:B]
            had_actor = True
        else:
            self.nt.zero_grad()
            had_actor = False
        self._allreduce_grad()
        self.nt.apply_grad(3)
        self.nt.sync()
        if had_actor:
            self._last_actor_loss = float(self.nt.loss[1])
        if float(self.nt.loss[3]) > 0:
            self.solver_iter += 1; self.actor_iter += 1
        succ = n_critic > 0
        if self.EnableTargetNet() and self.iter > 0 and self.iter % self.freeze_target_iters == 0:
            self.nt.update_target()
        return succ

    def UpdateActorBatchBuffer(self):
        ids = self.FetchActorMinibatch(self.batch) if self.stage_train else []
        n = len(ids)
        if n:
            self.nt.idx[self.batch:self.batch + n] = ids
            self.nt.actor_filter(n)
        self.nt.sync()
        if n:
            better = self.nt.better[:n].copy()
            self.actor_batch_buffer += [t for t, b in zip(ids, better) if b]


class _LossOnDemand:
    """The loss of the last native iteration is read from page-locked memory when somebody asks, not after every step (no host wait inside Train())."""

    @property
    def last_loss(self):
        if getattr(self, "_loss_pending", False):
            self.nt.sync(); self._last_loss = float(self.nt.loss[0]); self._loss_pending = False
        return None if self._last_loss is None else float(self._last_loss)

    def _settle(self):
        """a queued iteration must have read its index window before the host rewrites it"""
        if getattr(self, "_loss_pending", False):
            self.nt.sync(); self._last_loss = float(self.nt.loss[0]); self._loss_pending = False


class HipQNetTrainer(_LossOnDemand, _HipNetSide, QNetTrainer):
    """cQNetTrainer (trainer.QNetTrainer: minibatch draw) with the single-head net on the native step. Round 4: the whole iteration -- gather s', forward, new_q,
    gather s, forward, labels (own outputs with the taken action's entry replaced), backward, SGD -- is ONE recorded launch sequence (dtrl_trainer_value_step, kind 0);
    the host's part is the minibatch's slot indices in page-locked memory. native_targets=False keeps round 3's form (two dtrl_trainer_eval passes, the targets as
    framework ops, dtrl_trainer_step)."""

    native_targets = True

    def _q_problem(self, ids):
        self._order_staged()                    # replay rows stored from the staging area on the trainer's stream are read by framework ops here
        return super()._q_problem(ids)

    def Step(self):
        if not self.native_targets:
            return super().Step()
        if getattr(self, "_replay_dirty", False):
            self._after_torch(); self._replay_dirty = False
        ids = self.FetchMinibatch(self.batch)
        if len(ids) >= self.batch:
            self._settle()
            self.nt.idx[:self.batch] = ids
            self.nt.value_step(0)
            self.solver_iter += 1
            self._loss_pending = True
        return True


class HipCaclaTrainer(_LossOnDemand, _HipNetSide, CaclaTrainer):
    """cCaclaTrainer (trainer.CaclaTrainer: critic / actor schedule, off-policy candidates) with BOTH single-head nets on the native step: this object's net is the
    critic, self.actor is a HipQNetTrainer holding the actor net. Round 4: critic iteration = dtrl_trainer_value_step (kind 1), the candidates' TD test =
    dtrl_trainer_td_filter (mask and TD errors read from page-locked memory after the one host wait of a Train()), actor iteration = dtrl_trainer_action_step on the
    actor's trainer, which is bound to the critic's replay memory and runs on the critic's stream (one queue: stores, critic, filter, actor stay in program order)."""

    native_targets = True

    def Reset(self):
        super().Reset()
        net_file, solver_file, S, A, kw = self._actor_args
        kw = dict(kw); kw.pop("dtype", None); kw.pop("use_graphs", None)
        self.actor = HipQNetTrainer(net_file, solver_file, S, A, lib_path=self._trainer_lib, **kw)
        self.actor_batch = self.actor.batch
        self._actor_bound = False

    def _bind_actor(self):
        """the actor's trainer reads the tuples' states and actions from THIS trainer's replay memory, on this trainer's stream"""
        if not self._actor_bound:
            self.actor.nt.sync()
            self.actor.nt.bind_replay(self.mem.data_ptr(), self.flags_dev.data_ptr(), self.W)
            if self._stream is not None:
                self.actor._stream = self._stream; self.actor.nt.set_stream(self._stream.cuda_stream)
            self._actor_bound = True

    def UseStream(self, stream_ptr):
        super().UseStream(stream_ptr)
        if self.device.type == "cuda":
            self.actor.UseStream(stream_ptr)

    def Step(self):
        if not self.native_targets:
            return super().Step()
        if getattr(self, "_replay_dirty", False):
            self._after_torch(); self._replay_dirty = False
        self._bind_actor()
        ids = self.FetchMinibatch(self.batch)
        if len(ids) >= self.batch:
            self._settle()
            self.nt.idx[:self.batch] = ids
            self.nt.value_step(1)
            self.solver_iter += 1
            self._loss_pending = True
        self.UpdateActor()
        if self.EnableTargetNet() and self.iter > 0 and self.iter % self.freeze_target_iters == 0:
            self.nt.update_target()
        return True

    def UpdateActorBatchBuffer(self):
        if not self.native_targets:
            return super().UpdateActorBatchBuffer()
        ids = self._draw_actor_candidates()
        if not ids:
            return
        k = len(ids)
        self.nt.idx[self.batch:self.batch + k] = ids        # (the previous filter was waited for, so its window is free)
        self.nt.td_filter(k)
        self.nt.sync()                                       # the one host wait of a Train(): the mask decides what enters the actor batch buffer
        if getattr(self, "_loss_pending", False):
            self._last_loss = float(self.nt.loss[0]); self._loss_pending = False
        for t, b, d in zip(ids, self.nt.better[:k].copy(), self.nt.td[:k].copy()):
            if b:
                self.actor_batch_buffer.append(t); self.actor_batch_td.append(float(d))

    def UpdateActor(self):
        if not self.native_targets:
            return super().UpdateActor()
        if self.stage_train:
            self.UpdateActorBatchBuffer()
        B = self.actor_batch
        for _ in range(len(self.actor_batch_buffer) // B):
            self.actor._settle()
            ant = self.actor.nt
            ant.idx[ant.max_eval:ant.max_eval + B] = self.actor_batch_buffer[:B]
            ant.action_step()
            self.actor._loss_pending = True
            self.actor.solver_iter += 1
            self.actor_iter += 1
            del self.actor_batch_buffer[:B]; del self.actor_batch_td[:B]

    # the interface the training loop uses speaks for the ACTOR (what the rollout engine runs), as in trainer.CaclaTrainer
    def GetWeights(self): return self.actor.GetWeights()
    def SetWeights(self, w): self.actor.SetWeights(w)
    def WeightsDevicePtr(self): return self.actor.WeightsDevicePtr()
    @property
    def policy_nt(self): return self.actor.nt

    def GetCriticWeights(self):
        return self.nt.get_params(0)

    def SetCriticWeights(self, w):
        self.nt.set_params(0, np.asarray(w, np.float32)); self.nt.update_target()
