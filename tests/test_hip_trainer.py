"""The MI355X-native MACE trainer step (include/dtrl_trainer.h, deepterrainrl_amd/csrc/dtrl_trainer*.{h,hip}, deepterrainrl_amd/hip_trainer.py).

CPU tests run the SAME operand definitions and sequencing (dtrl_trainer_ops.h / dtrl_trainer_core.h) under plain host loops (tests/emul/libdtrl_trainer_emul.so,
tests only) against (i) the torch peer trainer.MACETrainer in float32 -- forward, one solver step, the fused critic / actor calls -- and (ii) the whole-trainer
numpy fp64 oracle oracle/trainer_ref.py over six iterations. The -m gpu twins run the HIP kernels of lib/libdtrl.so against the same oracle on the MI355X."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, REFDATA
import test_trainer as TT

EMUL_TRAINER_LIB = os.path.join(REPO, "tests", "emul", "libdtrl_trainer_emul.so")
S, A, NF, FS = TT.S, TT.A, TT.NF, TT.FS


def make_native(lib, device="cpu", **kw):
    from deepterrainrl_amd import hip_trainer as ht
    args = dict(mem_size=256, num_init_samples=100, freeze_target_iters=0, device=device, seed=21)
    args.update(kw)
    return ht.HipMACETrainer(TT.TRAIN, TT.SOLVER, S, A, lib_path=lib, **args)


def make_peer(device="cpu", **kw):
    args = dict(mem_size=256, num_init_samples=100, freeze_target_iters=0, device=device, dtype=torch.float32, seed=21, use_graphs=False)
    args.update(kw)
    return TT.make_trainer(**args)


def test_trainer_abi_symbols_are_exported():
    """every symbol include/dtrl_trainer.h declares is exported by the HIP library (and by the check build)"""
    import ctypes
    import re
    from deepterrainrl_amd import hip_trainer as ht, LIB_PATH
    hdr = open(os.path.join(REPO, "include", "dtrl_trainer.h")).read()
    declared = sorted(set(re.findall(r"\b(dtrl_trainer_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(ht.TRAINER_ABI_SYMBOLS)
    for path in (LIB_PATH, EMUL_TRAINER_LIB):
        L = ctypes.CDLL(path)
        for s in declared:
            assert hasattr(L, s), (path, s)


def run_forward_and_step_vs_torch_peer(lib, device, tol):
    """dtrl_trainer_eval / dtrl_trainer_step vs the torch net: outputs of both nets on 1, 32 and 64 rows, then one and two solver steps (gradient of every
    blob through the update, momentum, weight decay with the per-blob multipliers) from the same weights, normalisers and history."""
    rng = np.random.RandomState(5)
    t = make_native(lib, device)
    p = make_peer(device)
    w0 = p.GetWeights().copy()       # (a float32 CPU net hands out a view of its own storage)
    t.SetWeights(w0)
    io = rng.normal(0, 0.3, S); isc = rng.uniform(0.5, 2.0, S); oo = rng.normal(0, 0.2, 90); osc = rng.uniform(0.5, 3.0, 90)
    for x in (t, p):
        x.SetInputOffsetScale(io, isc); x.SetOutputOffsetScale(oo, osc)
    for n in (1, 32, 64):
        X = torch.as_tensor(rng.normal(0, 1, (n, S)).astype(np.float32), device=device)
        ya = t._eval(t.net, X); yb = p._eval(p.net, X)
        if device != "cpu":
            t.nt.sync()
        d = (ya - yb).abs().max().item()
        assert d < tol * max(1.0, yb.abs().max().item()), (n, d)
    X = torch.as_tensor(rng.normal(0, 1, (32, S)).astype(np.float32), device=device)
    Y = torch.as_tensor(rng.normal(0, 1, (32, 90)).astype(np.float32), device=device)
    for k in range(2):
        la = t._solver_step(X, Y); lb = p._solver_step(X, Y)
        assert abs(float(la) - float(lb)) < 10 * tol * max(1.0, abs(float(lb))), (k, float(la), float(lb))
        wa, wb = t.GetWeights().astype(np.float64), p.GetWeights().astype(np.float64)
        dw = np.abs(wa - wb).max()
        assert dw < tol * np.abs(wb).max() and np.abs(wb - w0).max() > 1e-5, (k, dw)
        ha, hb = t.nt.get_params(2), p.hflat.detach().cpu().numpy()
        assert np.abs(ha - hb).max() < tol * max(np.abs(hb).max(), 1e-3), k
    # per-blob agreement (a wrong transpose in one small blob would hide behind the largest weight above)
    off = 0
    for b in p.net.blobs():
        k = b.numel()
        da = np.abs(wa[off:off + k] - wb[off:off + k]).max(); mv = np.abs(wb[off:off + k] - w0[off:off + k]).max()
        assert da <= 50 * tol * max(mv, 1e-6) + 1e-7, (off, k, da, mv)
        off += k


def test_forward_and_step_vs_torch_peer():
    run_forward_and_step_vs_torch_peer(EMUL_TRAINER_LIB, "cpu", 2e-5)


def run_iterations_vs_numpy_oracle(om, lib, device, freeze, tol):
    """cMACETrainer iterations through the fused native calls (critic step, actor filter, actor step) vs oracle/trainer_ref.py (numpy fp64): same
    minibatches, critic targets, candidate decisions, labels, SGD -> same iteration counts, actor batch buffer and weights."""
    rng = np.random.RandomState(9)
    rows, flags = TT.random_rows(rng, 200, p_actor=0.5)
    t = make_native(lib, device, freeze_target_iters=freeze)
    r = TT.make_ref_trainer(om, t, 21, **({"freeze_target_iters": freeze} if freeze else {}))
    w0 = t.GetWeights().copy()
    r.set_weights(w0); t.SetWeights(w0)
    t.AddTuples(rows, flags); r.add_tuples(rows, flags)
    for k in range(6):
        t.Train(); r.train()
        assert (t.GetIter(), t.actor_iter) == (r.iter, r.actor_iter) and t.actor_batch_buffer == r.book.actor_batch, k
        assert abs(t.last_loss - r.last_loss) < 1e-3 * max(1.0, abs(r.last_loss)), (k, t.last_loss, r.last_loss)
    a = t.GetWeights().astype(np.float64)
    assert r.iter == 6 and r.actor_iter >= 1
    assert np.abs(a - r.w).max() < tol * np.abs(r.w).max() and np.abs(a - w0).max() > 1e-4, np.abs(a - r.w).max() / np.abs(r.w).max()
    io, isc, _, _ = t.GetOffsetScale()
    assert np.allclose(io, r.in_off, atol=1e-5) and np.allclose(isc, r.in_scale, rtol=1e-4)
    return t


@pytest.mark.parametrize("freeze", [0, 2])
def test_iterations_vs_numpy_oracle(om, freeze):
    run_iterations_vs_numpy_oracle(om, EMUL_TRAINER_LIB, "cpu", freeze, 2e-4)


def test_train_loop_runs_on_the_native_trainer(da):
    """train_loop.train(trainer="hip") end to end on the CPU check builds: rollouts (lane-loop engine) -> tuples -> native trainer step -> weights back."""
    from conftest import EmulScenario
    from deepterrainrl_amd import train_loop
    extra = {"terrain_seed": 3, "trainer_num_init_samples": 30, "trainer_replay_mem_size": 512, "trainer_freeze_target_iters": 4,
             "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"}
    st = train_loop.train("args/opt_args_train_mace.txt", REFDATA, num_envs=48, max_frames=70, trainer_device="cpu", scenario_cls=EmulScenario, extra_args=extra,
                          trainer="hip", trainer_lib=EMUL_TRAINER_LIB)
    assert st["iters"] >= 1 and np.all(np.isfinite(st["weights"])) and st["tuples"] >= 32


def run_staged_add_equals_plain_add(lib, device):
    """dtrl_trainer_add_staged (rows stored from the page-locked staging area on the trainer's stream) vs the framework-side store: same replay memory, flag
    words and training results -- through a ring wrap (300 rows into 256 slots) and a chunk holding a rejected (non-finite) row, which takes the plain path."""
    rng = np.random.RandomState(4)
    rows, flags = TT.random_rows(rng, 300, p_actor=0.5)
    rows[70, 5] = np.nan
    a = make_native(lib, device); b = make_native(lib, device)
    b.SetWeights(a.GetWeights())
    base = staged = 0
    for k in range(0, 300, 32):
        n = min(32, 300 - k)
        if k + n > base + staged:
            base = k; staged = a.StageTuples(rows[k:], flags[k:])
        sa = a.AddTuples(rows[k:k + n], flags[k:k + n], staged=k - base)
        sb = b.AddTuples(rows[k:k + n], flags[k:k + n])
        assert np.array_equal(sa, sb)
        a.Train(); b.Train()
    a.nt.sync(); b.nt.sync()
    if device != "cpu":
        torch.cuda.synchronize()
    ma, mb = a.mem.cpu().numpy(), b.mem.cpu().numpy()
    assert np.array_equal(ma, mb) and np.array_equal(a.flags_dev.cpu().numpy(), b.flags_dev.cpu().numpy()) and np.abs(ma).max() > 0
    assert (a.GetIter(), a.actor_iter, a.actor_batch_buffer) == (b.GetIter(), b.actor_iter, b.actor_batch_buffer) and a.GetIter() >= 5
    assert np.array_equal(a.GetWeights(), b.GetWeights())


def test_staged_add_equals_plain_add():
    run_staged_add_equals_plain_add(EMUL_TRAINER_LIB, "cpu")


def run_q_trainer_on_the_native_step(om, lib, device):
    """cQNetTrainer's iterations with the single-head Q net on dtrl_trainer_eval / dtrl_trainer_step (hip_trainer.HipQNetTrainer; rows stored from the staging area)
    vs oracle/trainer_ref.py's RefQTrainer: same minibatches, targets and SGD over six iterations through a ring wrap."""
    from deepterrainrl_amd import hip_trainer as ht
    rng = np.random.RandomState(12)
    rows, flags = TT.q_rows(rng, 300)
    t = ht.HipQNetTrainer(TT.QTRAIN, TT.QSOLVER, S, TT.QA, lib_path=lib, mem_size=256, num_init_samples=100, device=device, seed=8)
    r = TT.make_ref_q_trainer(om, t, 8)
    w0 = t.GetWeights(); t.SetWeights(w0); r.w = w0.astype(np.float64)
    k = 0
    while k < 300:
        st = t.StageTuples(rows[k:], flags[k:])
        for j in range(0, st, 50):
            t.AddTuples(rows[k + j:k + j + 50], flags[k + j:k + j + 50], staged=j)
        k += st
    r.add_tuples(rows, flags)
    for k in range(6):
        t.Train(); r.train()
        assert t.GetIter() == r.iter == k + 1
    a = t.GetWeights().astype(np.float64)
    assert np.abs(a - r.w).max() < 2e-4 * np.abs(r.w).max() and np.abs(a - w0).max() > 1e-4
    assert abs(t.last_loss - r.last_loss) < 1e-3 * max(1.0, abs(r.last_loss))
    return t


def test_q_trainer_on_the_native_step(om):
    run_q_trainer_on_the_native_step(om, EMUL_TRAINER_LIB, "cpu")


def cacla_factory(lib):
    def make(device, seed):
        from deepterrainrl_amd import hip_trainer as ht
        return ht.HipCaclaTrainer(TT.CRITIC[0], TT.CRITIC[1], TT.ACTOR[0], TT.ACTOR[1], S, TT.CA, lib_path=lib, mem_size=256, num_init_samples=100, freeze_target_iters=3, device=device, seed=seed)
    return make


def test_cacla_trainer_on_the_native_step(om):
    """cCaclaTrainer with critic AND actor on the native step (hip_trainer.HipCaclaTrainer) vs RefCaclaTrainer: 14 iterations, mid-training arrivals, target refresh"""
    TT.run_cacla_trainer_vs_restatement(om, "cpu", torch.float32, 3e-4, factory=cacla_factory(EMUL_TRAINER_LIB))


@pytest.mark.parametrize("arg,nparams", [("args/opt_args_train_q.txt", 461208), ("args/opt_args_train_cacla.txt", 463917)])
def test_q_and_cacla_train_loops_run_on_the_native_trainer(da, arg, nparams):
    """the Q and CACLA training configurations end to end with trainer="hip" (check builds): rollouts -> tuples -> native eval / step -> policy hand-over from the
    trainer's device buffer (CACLA: the actor's)"""
    from conftest import EmulScenario
    from deepterrainrl_amd import train_loop
    st = train_loop.train(arg, REFDATA, num_envs=48, max_frames=55, trainer_device="cpu", scenario_cls=EmulScenario, trainer="hip", trainer_lib=EMUL_TRAINER_LIB,
                          extra_args={"terrain_seed": 3, "trainer_num_init_samples": 60, "trainer_replay_mem_size": 512, "trainer_init_input_offset_scale": "false"})
    assert st["frames"] == 55 and st["tuples"] >= 60 and st["iters"] >= 2
    assert np.all(np.isfinite(st["weights"])) and st["weights"].size == nparams


# ---- the HIP kernels on the MI355X ----
@pytest.mark.gpu
def test_gpu_q_and_cacla_trainers_on_the_native_step(om):
    t = run_q_trainer_on_the_native_step(om, None, "cuda")
    assert "libdtrl.so" in open("/proc/self/maps").read() and t.mem.is_cuda
    TT.run_cacla_trainer_vs_restatement(om, "cuda", torch.float32, 3e-4, factory=cacla_factory(None))



@pytest.mark.gpu
def test_gpu_staged_add_equals_plain_add():
    run_staged_add_equals_plain_add(None, "cuda")



@pytest.mark.gpu
def test_gpu_framework_reads_wait_for_staged_stores():
    """rows stored from the staging area are written on the TRAINER's stream; a framework read of the replay memory on torch's current stream
    (the normaliser statistics of UpdateOffsetScale, a CACLA / Q minibatch) must wait for them. The trainer's stream is held up by a long spin kernel in front
    of the stores, as a frame kernel beside the trainer does."""
    from deepterrainrl_amd import hip_trainer as ht
    rng = np.random.RandomState(3)
    rows, flags = TT.q_rows(rng, 120)
    t = ht.HipQNetTrainer(TT.QTRAIN, TT.QSOLVER, S, TT.QA, lib_path=None, mem_size=256, num_init_samples=100000, device="cuda", seed=8)
    torch.cuda.synchronize()
    with torch.cuda.stream(t._stream):
        torch.cuda._sleep(int(2e8))            # ~100 ms in front of the stores on the trainer's stream
    n = t.StageTuples(rows, flags)
    assert n == 120
    t.AddTuples(rows, flags, staged=0)
    t.UpdateOffsetScale()
    io, isc, _, _ = t.GetOffsetScale()
    X = rows[:, 1:1 + S].astype(np.float64)
    std = X.std(0)
    assert np.allclose(np.asarray(io), -X.mean(0), atol=1e-6)
    assert np.allclose(np.asarray(isc), np.where(std == 0, 0, 1.0 / np.where(std == 0, 1, std)), rtol=1e-5)
    rows2, flags2 = TT.q_rows(rng, 64)
    assert t.StageTuples(rows2, flags2) == 64      # (waits for the previous stores; the spin is then queued in front of the new ones)
    with torch.cuda.stream(t._stream):
        torch.cuda._sleep(int(2e8))
    slots = t.AddTuples(rows2, flags2, staged=0)
    got = t._rows(list(slots)).cpu().numpy()
    assert np.array_equal(got, rows2.astype(np.float32))


@pytest.mark.gpu
def test_gpu_forward_and_step_vs_torch_peer():
    run_forward_and_step_vs_torch_peer(None, "cuda", 5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("freeze", [0, 2])
def test_gpu_iterations_vs_numpy_oracle(om, freeze):
    t = run_iterations_vs_numpy_oracle(om, None, "cuda", freeze, 2e-4)
    maps = open("/proc/self/maps").read()
    assert "libdtrl.so" in maps and t.mem.is_cuda


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 2, 3, 4])
def test_gpu_every_fused_mode_of_the_native_step(om, monkeypatch, mode):
    """DTRL_TRAINER_FUSED selects how the native step is cut into launches (0 layer by layer, 1 fused forward = the default the other tests run, 2 fused forward and
    backward, 3 forward split into conv stack / split-K GEMM / FC chain, 4 fused forward + fused FC data-gradient chain with the weight gradients on a second graph
    branch). The non-default cuts are kept as measured alternatives (profiles/r06_trainer.txt): each must compute the same step -- forward and two solver steps against
    the PyTorch peer, six cMACETrainer iterations against the numpy restatement."""
    monkeypatch.setenv("DTRL_TRAINER_FUSED", str(mode))
    run_forward_and_step_vs_torch_peer(None, "cuda", 2e-4)
    run_iterations_vs_numpy_oracle(om, None, "cuda", 2, 2e-4)


@pytest.mark.gpu
def test_gpu_equals_the_plain_loop_build(om):
    """HIP kernels vs the plain-loop build of the same operand definitions: six iterations, identical decisions, weights within fp32 contraction noise."""
    rng = np.random.RandomState(9)
    rows, flags = TT.random_rows(rng, 200, p_actor=0.5)
    a = make_native(None, "cuda"); b = make_native(EMUL_TRAINER_LIB, "cpu")
    b.SetWeights(a.GetWeights())
    for t in (a, b):
        t.AddTuples(rows, flags)
    for k in range(6):
        a.Train(); b.Train()
        assert (a.GetIter(), a.actor_iter, a.actor_batch_buffer) == (b.GetIter(), b.actor_iter, b.actor_batch_buffer), k
    wa, wb = a.GetWeights().astype(np.float64), b.GetWeights().astype(np.float64)
    assert np.abs(wa - wb).max() < 2e-5 * np.abs(wb).max()


@pytest.mark.gpu
def test_gpu_overlapped_training_loop_is_reproducible():
    """train_loop.train(trainer="hip", overlap=True) on the MI355X -- frames relaunched before the drain, tuple rings in host memory, tuples staged and stored on
    the trainer's stream, weights parked for the next launch on the trainer's stream -- twice from the same seeds: every hand-over is host-ordered and the drained
    rows are sorted by env id, so iteration counts, tuple counts and every weight must come out identical (a race in any of those pieces would show here)."""
    from deepterrainrl_amd import train_loop
    extra = {"terrain_seed": 3, "trainer_num_init_samples": 1500, "trainer_replay_mem_size": 20000, "trainer_freeze_target_iters": 50,
             "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"}
    runs = []
    for k in range(2):
        st = train_loop.train("args/opt_args_train_mace.txt", REFDATA, num_envs=1536, max_frames=70, trainer_device="cuda", extra_args=dict(extra), trainer="hip", overlap=True, seed=5)
        runs.append(st)
        assert st["frames"] == 70 and st["iters"] >= 20 and np.all(np.isfinite(st["weights"]))
    a, b = runs
    assert (a["iters"], a["tuples"]) == (b["iters"], b["tuples"]) and a["tuples"] >= 1500
    assert np.array_equal(a["weights"], b["weights"])
    assert "libdtrl.so" in open("/proc/self/maps").read()


class _ReplayDraws:
    """numpy RandomState's randint(lo, hi) call shape over a recorded list of draws (the reference trainer's cMathUtil::gRand stream, frozen)"""

    def __init__(self, draws):
        self.draws, self.k = [int(x) for x in draws], 0

    def randint(self, lo, hi=None, size=None):
        if hi is None:
            lo, hi = 0, lo
        if size is not None:
            return np.array([self.randint(lo, hi) for _ in range(size)], np.int64)
        v = self.draws[self.k]; self.k += 1
        assert lo <= v < hi, "the product asked for a draw the reference did not make at this point (%d not in [%d, %d))" % (v, lo, hi)
        return v


def run_native_trainer_vs_frozen_reference_trainer(om, lib, device, tol):
    """The native step against a run of the REFERENCE'S OWN cMACETrainer frozen on the CPU box (tests/golden/make_ref_golden_learn.py: /root/reference/learning compiled
    unchanged; frozen target refreshed every 2 iterations, tuples arriving in three batches through a ring wrap): the product replays the reference's index draws and
    must land on its iteration counters, stage and three index buffers after every Train(), consume exactly the draws the reference consumed, and end at its weights."""
    g = np.load(os.path.join(REPO, "tests", "golden", "ref_golden_learn.npz"))
    rng = np.random.RandomState(9)
    rows, flags = TT.random_rows(rng, 420, p_actor=0.5)
    d = om.parse_deploy_prototxt(TT.DEPLOY)
    w0 = om.xavier_weights(d, int(g["w_seed"]))
    t = make_native(lib, device, mem_size=256, num_init_samples=100, seed=21, freeze_target_iters=int(g["freeze"]))
    t.rng = _ReplayDraws(g["draws"])
    t.SetWeights(w0)
    k = i = 0
    for n_new, n_train in g["schedule"]:
        t.AddTuples(rows[k:k + n_new], flags[k:k + n_new]); k += int(n_new)
        for _ in range(int(n_train)):
            t.Train()
            assert [t.GetIter(), t.actor_iter, int(t.stage_train)] == g["s%d_counters" % i].tolist(), i
            assert list(t.critic_buffer) == g["s%d_critic" % i].tolist() and list(t.actor_buffer) == g["s%d_actor" % i].tolist(), i
            assert list(t.actor_batch_buffer) == g["s%d_actor_batch" % i].tolist(), i
            i += 1
    assert i == int(g["n_steps"]) and t.rng.k == len(t.rng.draws)
    w = t.GetWeights().astype(np.float64)
    scale = float(g["w_norm"]) / np.sqrt(w.size)
    assert np.abs(w[g["pick"]] - g["w_pick"]).max() < tol * np.abs(g["w_pick"]).max(), np.abs(w[g["pick"]] - g["w_pick"]).max()
    assert abs(np.linalg.norm(w) - float(g["w_norm"])) < tol * float(g["w_norm"]) and scale > 0
    io, isc, _, _ = t.GetOffsetScale()
    assert np.allclose(io, g["in_off"], rtol=0, atol=1e-6) and np.allclose(isc, g["in_scale"], rtol=1e-5)
    return t


def test_native_trainer_vs_frozen_reference_trainer(om):
    run_native_trainer_vs_frozen_reference_trainer(om, EMUL_TRAINER_LIB, "cpu", 3e-4)


@pytest.mark.gpu
def test_gpu_native_trainer_vs_frozen_reference_trainer(om):
    t = run_native_trainer_vs_frozen_reference_trainer(om, None, "cuda", 3e-4)
    assert "libdtrl.so" in open("/proc/self/maps").read() and t.mem.is_cuda


@pytest.mark.gpu
def test_gpu_native_trainers_vs_the_compiled_reference_trainers(om):
    """On the GPU box the compiled reference trainers travel prebuilt (oracle/_ref/libref_learn.so): the HIP MACE trainer in lock-step with cMACETrainer itself"""
    from oracle import reflearn
    if not reflearn.available():
        pytest.skip("oracle/_ref/libref_learn.so not shipped")
    import test_reference_learn as TR
    TR.run_reference_vs_product_mace(reflearn, om, make_native(None, "cuda", mem_size=256, num_init_samples=100, seed=21, freeze_target_iters=2), 3e-4, 2)


def test_data_parallel_step_two_halves_equal_one_batch_of_64():
    """include/dtrl_trainer.h, data-parallel step: two trainers (batch 32) run dtrl_trainer_grad_step on the two halves of a batch, their gradient buffers are
    summed (what the all-reduce does, sample counts included) and dtrl_trainer_apply_grad updates both -- the same weights as ONE trainer stepping the concatenated
    batch of 64 with dtrl_trainer_step, over three iterations with momentum; a 'rank' that contributes nothing (dtrl_trainer_zero_grad) leaves the mean over the
    other's rows; no rows anywhere = no update."""
    import ctypes as C
    rng = np.random.RandomState(3)
    one = make_native(EMUL_TRAINER_LIB, "cpu")
    w0 = one.GetWeights().copy()
    from deepterrainrl_amd import hip_trainer as ht
    big = ht.HipMACETrainer(TT.TRAIN, TT.SOLVER, S, A, lib_path=EMUL_TRAINER_LIB, mem_size=256, num_init_samples=100, freeze_target_iters=0, device="cpu", seed=21)
    # a batch-64 trainer: the same net with the solver's batch doubled
    from deepterrainrl_amd.hip_trainer import NativeTrainer, desc_from_net
    cdesc = desc_from_net(big.desc, S, 64, 192, big.solver, big.discount, False)
    nt64 = NativeTrainer(cdesc, -1, EMUL_TRAINER_LIB)
    halves = [make_native(EMUL_TRAINER_LIB, "cpu") for _ in range(2)]
    io = rng.normal(0, 0.3, S); isc = rng.uniform(0.5, 2.0, S); oo = rng.normal(0, 0.2, 90); osc = rng.uniform(0.5, 3.0, 90)
    nt64.set_params(0, w0); nt64.set_params(3, big.rate_mult.numpy()); nt64.set_params(4, big.decay_mult.numpy()); nt64.set_normalizers(io, isc, oo, osc)
    for h in halves:
        h.SetWeights(w0); h.SetInputOffsetScale(io, isc); h.SetOutputOffsetScale(oo, osc)
    P = nt64.num_params

    def gbuf(nt):
        return np.ctypeslib.as_array((C.c_float * (P + 1)).from_address(nt.grad_device()))
    for it in range(3):
        X = rng.normal(0, 1, (64, S)).astype(np.float32); Y = rng.normal(0, 1, (64, 90)).astype(np.float32)
        nt64.step(X.ctypes.data, Y.ctypes.data); nt64.sync()
        for k, h in enumerate(halves):
            xs = np.ascontiguousarray(X[32 * k:32 * k + 32]); ys = np.ascontiguousarray(Y[32 * k:32 * k + 32])
            h.nt.grad_step(xs.ctypes.data, ys.ctypes.data); h.nt.sync()
        total = gbuf(halves[0].nt) + gbuf(halves[1].nt)
        assert total[P] == 64
        for h in halves:
            gbuf(h.nt)[:] = total
            h.nt.apply_grad(2); h.nt.sync()
            assert h.nt.loss[2] == 64
    a = nt64.get_params(0).astype(np.float64)
    for h in halves:
        b = h.GetWeights().astype(np.float64)
        assert np.abs(a - b).max() < 2e-6 * np.abs(a).max() and np.abs(a - w0).max() > 1e-4
    assert np.array_equal(halves[0].GetWeights(), halves[1].GetWeights())
    # one rank without a batch: the update is the mean over the other's 32 rows = that rank's plain step
    solo = make_native(EMUL_TRAINER_LIB, "cpu"); solo.SetWeights(w0); solo.SetInputOffsetScale(io, isc); solo.SetOutputOffsetScale(oo, osc)
    pair = [make_native(EMUL_TRAINER_LIB, "cpu") for _ in range(2)]
    for h in pair:
        h.SetWeights(w0); h.SetInputOffsetScale(io, isc); h.SetOutputOffsetScale(oo, osc)
    X = rng.normal(0, 1, (32, S)).astype(np.float32); Y = rng.normal(0, 1, (32, 90)).astype(np.float32)
    solo.nt.step(X.ctypes.data, Y.ctypes.data); solo.nt.sync()
    pair[0].nt.grad_step(X.ctypes.data, Y.ctypes.data); pair[1].nt.zero_grad(); pair[0].nt.sync(); pair[1].nt.sync()
    total = gbuf(pair[0].nt) + gbuf(pair[1].nt)
    assert total[P] == 32
    for h in pair:
        gbuf(h.nt)[:] = total; h.nt.apply_grad(2); h.nt.sync()
        assert np.array_equal(h.GetWeights(), solo.GetWeights())
    w1 = pair[1].GetWeights().copy()
    pair[1].nt.zero_grad(); pair[1].nt.apply_grad(3); pair[1].nt.sync()
    assert pair[1].nt.loss[3] == 0 and np.array_equal(pair[1].GetWeights(), w1)       # nobody had a batch: nothing moves (the history neither)


def test_data_parallel_trainer_on_one_rank_equals_the_plain_native_trainer(om):
    """HipMACETrainerDP without a process group: gradient / apply in two calls instead of the fused step -- the same weights, bit for bit, as HipMACETrainer"""
    from deepterrainrl_amd import hip_trainer as ht
    rng = np.random.RandomState(9)
    rows, flags = TT.random_rows(rng, 200, p_actor=0.5)
    kw = dict(lib_path=EMUL_TRAINER_LIB, mem_size=256, num_init_samples=100, freeze_target_iters=0, device="cpu", seed=21)
    a = ht.HipMACETrainer(TT.TRAIN, TT.SOLVER, S, A, **kw)
    b = ht.HipMACETrainerDP(TT.TRAIN, TT.SOLVER, S, A, dist=None, **kw)
    b.SetWeights(a.GetWeights())
    a.AddTuples(rows, flags); b.AddTuples(rows, flags)
    for k in range(6):
        a.Train(); b.Train()
        assert (a.GetIter(), a.actor_iter, a.actor_batch_buffer) == (b.GetIter(), b.actor_iter, b.actor_batch_buffer), k
    assert a.GetIter() == 6 and a.actor_iter >= 1
    assert np.array_equal(a.GetWeights(), b.GetWeights())
    ioa, isa, _, _ = a.GetOffsetScale(); iob, isb, _, _ = b.GetOffsetScale()
    assert np.allclose(ioa, iob, atol=1e-6) and np.allclose(isa, isb, rtol=1e-5)


@pytest.mark.gpu
def test_gpu_batch_neural_net_shim_and_the_reference_trainer_on_the_hip_nets(om):
    """include/BatchNeuralNet.h on the HIP library: (i) the driver compiled inside the reference's header tree (tests/shim/drive_shim_net_hip, prebuilt) through
    cNeuralNet's calls; (ii) the reference's own cMACETrainer with its nets forwarded to the shim (oracle/_ref/libref_learn_native_hip.so, prebuilt) against
    HipMACETrainer on the MI355X."""
    import subprocess
    from oracle import reflearn
    exe = os.path.join(REPO, "tests", "shim", "drive_shim_net_hip")
    if not os.path.exists(exe) or not os.path.exists(reflearn.NATIVE_HIP_LIB_PATH) or not reflearn.available():
        pytest.skip("prebuilt shim driver / libref_learn_native_hip.so not shipped")
    r = subprocess.run([exe, REFDATA, "/tmp"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "shim net ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    import test_reference_learn as TR
    TR.run_reference_trainer_on_the_product_nets(reflearn, om, reflearn.NATIVE_HIP_LIB_PATH, make_native(None, "cuda", mem_size=256, num_init_samples=100, seed=21, freeze_target_iters=2), 3e-4)


DP_GPU_WORKER = r"""
import json, os, sys, numpy as np
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import torch, torch.distributed as dist
import test_trainer as TT
from conftest import REFDATA
from deepterrainrl_amd import hip_trainer as ht, train_loop
import deepterrainrl_amd as da
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda:0"))
S, A = TT.S, TT.A
rng = np.random.RandomState(9)
rows, flags = TT.random_rows(rng, 200, p_actor=0.5)
kw = dict(lib_path=None, mem_size=256, num_init_samples=100, freeze_target_iters=0, device="cuda", seed=21)
a = ht.HipMACETrainer(TT.TRAIN, TT.SOLVER, S, A, **kw)
b = ht.HipMACETrainerDP(TT.TRAIN, TT.SOLVER, S, A, dist=dist, **kw)
b.SetWeights(a.GetWeights())
a.AddTuples(rows, flags); b.AddTuples(rows, flags)
book = []
for k in range(6):
    a.Train(); b.Train()
    book.append([a.GetIter() == b.GetIter(), a.actor_iter == b.actor_iter, list(a.actor_batch_buffer) == list(b.actor_batch_buffer)])
wa, wb = a.GetWeights().astype(np.float64), b.GetWeights().astype(np.float64)
out = dict(book=book, iters=int(a.GetIter()), actor_iters=int(a.actor_iter), rel=float(np.abs(wa - wb).max() / np.abs(wa).max()), maps=("libdtrl.so" in open("/proc/self/maps").read()), rccl=str(torch.cuda.nccl.version()))
st = train_loop.train_distributed("args/opt_args_train_mace.txt", REFDATA, 256, dist, max_frames=70, trainer_device="cuda:0", local_device_id=0, trainer="hip", mode="data_parallel", overlap=True,
                                  extra_args={{"terrain_seed": 3, "trainer_num_init_samples": 120, "trainer_replay_mem_size": 2048, "trainer_freeze_target_iters": 4,
                                              "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"}})
out.update(loop_frames=int(st["frames"]), loop_iters=int(st["iters"]), loop_tuples=int(st["tuples"]), loop_drained=int(st["batch"].TupleStats()["drained"]),
           loop_finite=bool(np.all(np.isfinite(st["weights"]))), loop_hist=float(np.abs(st["trainer"].nt.get_params(2)).max()))
print("DPGPU " + json.dumps(out), flush=True)
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_gpu_data_parallel_trainer_and_loop_on_a_one_rank_rccl_group(tmp_path):
    """The data-parallel step on the HIP kernels with RCCL in the loop (one rank: the all-reduce is the identity, but it is issued -- on the trainer's stream, on the
    bound gradient tensor): HipMACETrainerDP (critic_grad / all_reduce / apply_grad, actor likewise) against HipMACETrainer's fused steps over six Train() calls -- the
    same decisions, the same weights to float32 rounding (the two update kernels are compiled separately, contraction may differ); then
    train_distributed(mode="data_parallel", overlap=True) end to end on the real engine: every tuple of the rank's own envs consumed, iterations made, finite weights."""
    import json
    script = tmp_path / "dp_gpu_worker.py"
    script.write_text(DP_GPU_WORKER.format(repo=REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", DTRL_FORCE_COLLECTIVES="1")
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("DPGPU ")][-1]
    d = json.loads(line[6:])
    assert d["maps"] and all(all(x) for x in d["book"]) and d["iters"] == 6 and d["actor_iters"] >= 1, d
    assert d["rel"] < 2e-6, d
    assert d["loop_frames"] == 70 and d["loop_iters"] >= 2 and d["loop_finite"] and d["loop_hist"] > 0 and d["loop_tuples"] == d["loop_drained"] >= 120, d


TWO_RANK_TRAIN_WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import torch, torch.distributed as dist
from conftest import REFDATA
from deepterrainrl_amd import train_loop
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
st = train_loop.train_distributed("args/opt_args_train_mace.txt", REFDATA, {envs}, dist, max_frames={frames}, device="cuda:0", trainer_device="cuda:0", local_device_id=0,
                                  trainer="hip", extra_args={extra!r}, seed=5, block_rows={envs})
assert "libdtrl.so" in open("/proc/self/maps").read()
if dist.get_rank() == 0:
    np.savez(os.path.join({out!r}, "dist_train.npz"), weights=st["weights"], iters=st["iters"], tuples=st["tuples"], in_off=st["offset_scale"][0], frames=st["frames"])
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_gpu_two_rank_training_sharing_the_gpu_equals_single_process(tmp_path):
    """cScenarioTrain's loop across two ranks ON THE HIP ENGINE AND THE HIP TRAINER (the box has one GPU and RCCL refuses two ranks on one device: a gloo group whose
    collectives are staged through pinned host memory): both processes roll out their half of the 192 global envs on cuda:0, every frame's tuples are packed on the
    device and gathered to rank 0, rank 0 runs the native MACE trainer step and broadcasts [weights | normalisers] back. The run must equal train() in ONE process on the
    same GPU bit for bit -- iterations, tuples, every weight -- as its CPU twin (tests/test_multi_gpu_gloo.py, lane-loop backends) does."""
    from deepterrainrl_amd import train_loop
    envs, frames = 192, 60
    extra = {"terrain_seed": 3, "trainer_num_init_samples": 300, "trainer_replay_mem_size": 4096, "trainer_freeze_target_iters": 8,
             "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"}
    script = tmp_path / "two_rank_train_worker.py"
    script.write_text(TWO_RANK_TRAIN_WORKER.format(repo=REPO, out=str(tmp_path), envs=envs, frames=frames, extra=extra))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29681", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = np.load(tmp_path / "dist_train.npz")
    st = train_loop.train("args/opt_args_train_mace.txt", REFDATA, num_envs=envs, max_frames=frames, trainer_device="cuda", extra_args=dict(extra), trainer="hip", seed=5)
    assert int(d["frames"]) == st["frames"] == frames
    assert int(d["iters"]) == int(st["iters"]) >= 5 and int(d["tuples"]) == int(st["tuples"]) >= 300, (int(d["iters"]), st["iters"], int(d["tuples"]), st["tuples"])
    assert np.all(np.isfinite(d["weights"])) and np.array_equal(d["weights"], st["weights"]) and np.array_equal(d["in_off"], st["offset_scale"][0])


TWO_RANK_DP_WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import torch, torch.distributed as dist
from conftest import REFDATA
from deepterrainrl_amd import train_loop
torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
st = train_loop.train_distributed("args/opt_args_train_mace.txt", REFDATA, {envs}, dist, max_frames={frames}, trainer_device="cuda:0", local_device_id=0, trainer="hip",
                                  mode="data_parallel", extra_args={extra!r})
assert "libdtrl.so" in open("/proc/self/maps").read()
t = st["trainer"]
t._order_staged()
X = t.mem[:t.num_tuples, 1:1 + t.S].to(torch.float64)
stats = torch.cat([torch.tensor([float(X.shape[0])], dtype=torch.float64, device=X.device), X.sum(0), (X * X).sum(0)]).cpu()
t.UpdateOffsetScale()          # the pooled normaliser (one all-reduce of count / sum / sum of squares, staged here): every rank calls it, after the training it did not feed
off, sc = t.GetOffsetScale()[:2]
np.savez(os.path.join({out!r}, "dp_rank%d.npz" % rank), weights=st["weights"], iters=st["iters"], actor_iters=st["actor_iters"], tuples=st["tuples"], frames=st["frames"],
         pooled_off=np.asarray(off), pooled_scale=np.asarray(sc), hist=t.nt.get_params(2), stats=stats.numpy(), drained=st["batch"].TupleStats()["drained"])
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_gpu_two_rank_data_parallel_training_sharing_the_gpu(tmp_path):
    """train_distributed(mode="data_parallel") on two ranks with the HIP engine and the HIP trainer's gradient / apply kernels on both (sharing the one MI355X; gloo
    group, the two gradient all-reduces per Train() and the pooled normaliser staged through the host): no tuple gather and no weight broadcast, yet both ranks end
    with bit-identical weights and solver history and the same counters; the input normaliser both carry is the one of the POOLED begin states; each rank trained on
    its own envs' tuples only. The GPU twin of tests/test_multi_gpu_gloo.py::test_two_rank_data_parallel_training."""
    envs, frames = 512, 70
    extra = {"terrain_seed": 3, "trainer_num_init_samples": 600, "trainer_replay_mem_size": 8192, "trainer_freeze_target_iters": 4,
             "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"}
    script = tmp_path / "two_rank_dp_worker.py"
    script.write_text(TWO_RANK_DP_WORKER.format(repo=REPO, out=str(tmp_path), envs=envs, frames=frames, extra=extra))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29683", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    a, b = np.load(tmp_path / "dp_rank0.npz"), np.load(tmp_path / "dp_rank1.npz")
    assert int(a["frames"]) == int(b["frames"]) == frames
    assert int(a["iters"]) == int(b["iters"]) >= 2 and int(a["actor_iters"]) == int(b["actor_iters"]), (int(a["iters"]), int(b["iters"]))
    assert np.array_equal(a["weights"], b["weights"]) and np.array_equal(a["hist"], b["hist"]) and np.all(np.isfinite(a["weights"]))
    assert np.abs(a["hist"]).max() > 0
    assert int(a["tuples"]) == int(a["drained"]) >= 300 and int(b["tuples"]) == int(b["drained"]) >= 300
    assert not np.array_equal(a["stats"], b["stats"])
    pooled = a["stats"] + b["stats"]
    S = (len(pooled) - 1) // 2
    n = pooled[0]; mean = pooled[1:1 + S] / n
    std = np.sqrt(np.maximum(pooled[1 + S:] / n - mean * mean, 0.0))
    exp_scale = np.where(std == 0, 0.0, 1.0 / np.where(std == 0, 1.0, std))
    for r in (a, b):
        assert np.allclose(r["pooled_off"], -mean, rtol=1e-6, atol=1e-9) and np.allclose(r["pooled_scale"], exp_scale, rtol=1e-5, atol=1e-9)


def test_create_from_files_rejects_a_net_with_a_missing_head_instead_of_crashing(tmp_path):
    """ParseTrainerFiles looked every a<f>_ip0 / a<f>_ip1 of val_ip1's fragment count up with std::map::operator[] -- a net with fewer actor heads (or a
    malformed prototxt) dereferenced a null layer inside dtrl_trainer_create_from_files / cBatchNeuralNet::LoadNet. Now: an error naming the layer."""
    import ctypes as C
    lib = C.CDLL(os.path.join(REPO, "tests", "emul", "libdtrl_trainer_emul.so"))
    lib.dtrl_trainer_create_from_files.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.dtrl_trainer_last_error.restype = C.c_char_p; lib.dtrl_trainer_last_error.argtypes = [C.c_void_p]
    lib.dtrl_trainer_destroy.argtypes = [C.c_void_p]
    src = open(os.path.join(REFDATA, "data/policies/dog/nets/dog_mace3_deploy.prototxt")).read()
    good = tmp_path / "good_deploy.prototxt"; good.write_text(src)
    h = C.c_void_p()
    assert lib.dtrl_trainer_create_from_files(str(good).encode(), b"", REFDATA.encode(), 0.9, 0, -1, C.byref(h)) == 0 and h.value
    lib.dtrl_trainer_destroy(h)
    cut = "\n".join(l for l in src.splitlines() if '"a2_ip1"' not in l)          # the third actor head's output layer removed; val_ip1 still announces 3 fragments
    assert cut != src and '"a2_ip1"' not in cut
    bad = tmp_path / "bad_deploy.prototxt"; bad.write_text(cut)
    h = C.c_void_p()
    rc = lib.dtrl_trainer_create_from_files(str(bad).encode(), b"", REFDATA.encode(), 0.9, 0, -1, C.byref(h))
    assert rc != 0 and not h.value
    assert b"a2_ip1" in lib.dtrl_trainer_last_error(None)
