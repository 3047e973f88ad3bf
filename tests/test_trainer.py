"""MACE trainer: product = deepterrainrl_amd/trainer.py (torch tensors on the device), checker = oracle/trainer_ref.py
(plain numpy restatement of the reference's bookkeeping and update rule) + the C++ oracle's network forward."""
import os

import numpy as np
import pytest
import torch

from conftest import REFDATA, dog_policy

NETS = os.path.join(REFDATA, "data/policies/dog/nets")
TRAIN, SOLVER, DEPLOY = (os.path.join(NETS, "dog_mace3_%s.prototxt" % k) for k in ("train", "solver", "deploy"))
S, A, NF, FS = 283, 30, 3, 29


def make_trainer(**kw):
    from deepterrainrl_amd import trainer as tr
    args = dict(mem_size=64, num_init_samples=40, freeze_target_iters=0, device="cpu", dtype=torch.float64, seed=3)
    args.update(kw)
    return tr.MACETrainer(TRAIN, SOLVER, S, A, **args)


def random_rows(rng, n, p_actor=0.4, p_fail=0.2):
    rows = rng.normal(0, 1, size=(n, 1 + 2 * S + A)).astype(np.float32)
    rows[:, 0] = rng.uniform(0, 1, n)
    rows[:, 1 + S] = rng.randint(0, NF, n)
    flags = (rng.uniform(size=n) < p_actor) * 4 + (rng.uniform(size=n) < p_fail) * 1 + (rng.uniform(size=n) < 0.3) * 2
    return rows, flags.astype(np.int64)


def test_prototxt_parsing_and_blob_order_vs_oracle_forward(om):
    from deepterrainrl_amd import trainer as tr
    d = tr.parse_net(TRAIN)
    assert d["batch_size"] == 32 and d["in_size"] == S and d["n_terrain"] == 200 and (d["n_frags"], d["frag_size"]) == (NF, FS)
    assert tr.parse_net(DEPLOY)["in_size"] == S
    s = tr.parse_solver(SOLVER)
    assert (s["base_lr"], s["momentum"], s["weight_decay"], s["lr_policy"]) == (0.001, 0.9, 0.0005, "fixed")
    t = make_trainer()
    assert t.net.num_params() == 570474 and len(t.net.blob_mults) == 2 * 13
    assert t.net.blob_mults[0] == (1.0, 1.0) and t.net.blob_mults[1] == (2.0, 1.0)          # conv: lr_mult only -> decay_mult defaults to 1
    assert t.net.blob_mults[6] == (1.0, 1.0) and t.net.blob_mults[7] == (2.0, 0.0)          # terr_ip0: bias not decayed
    # flat weights in the rollout engine's blob order: the torch net must reproduce the oracle's (C++, fp64) forward
    desc, w, io, isc, oo, osc = dog_policy(om)
    t.SetWeights(w); t.SetInputOffsetScale(io, isc); t.SetOutputOffsetScale(oo, osc)
    assert np.array_equal(t.GetWeights(), w)
    m, _ = om.build_model("args/dog_slopes_mixed_args.txt", REFDATA)
    e = om.OracleEnv(m, terrain_seed=1, policy=(desc, w, io, isc, oo, osc))
    rng = np.random.RandomState(0)
    for _ in range(3):
        x = rng.normal(0, 1, S) / np.where(isc == 0, 1, isc) - io
        y_ref = e.nn_eval(x)
        y = t.Eval(x)[0]
        assert np.abs(y - y_ref).max() < 1e-9 * max(1.0, np.abs(y_ref).max())


def test_bookkeeping_targets_and_labels_vs_oracle():
    from oracle import trainer_ref as ref
    t = make_trainer(mem_size=48, num_init_samples=10 ** 9)       # stays in the init stage: pure bookkeeping
    book = ref.RefTrainerBook(S, A, NF, FS, 48, t.batch, t.discount)
    rng = np.random.RandomState(11)
    rows, flags = random_rows(rng, 130)                            # wraps the 48-slot ring almost three times
    rows[17, 5] = np.nan; rows[40, 0] = np.inf                     # CheckTuple rejects these
    t.actor_batch_buffer = [3, 9, 3, 20]                           # stale candidates must be purged when their slot is overwritten
    book.actor_batch = [3, 9, 3, 20]
    for k in range(0, 130, 7):
        slots = t.AddTuples(rows[k:k + 7], flags[k:k + 7])
        exp = [book.add(rows[i], flags[i]) for i in range(k, min(k + 7, 130))]
        assert list(slots) == exp
        assert t.critic_buffer == book.critic and t.actor_buffer == book.actor and t.actor_batch_buffer == book.actor_batch
        assert (t.head, t.num_tuples) == (book.head, book.num)
    assert np.array_equal(t.mem.cpu().numpy(), book.mem) and np.array_equal(t.flags, book.flags)
    net_eval = lambda x: t.Eval(x)[0]
    ids = t.critic_buffer[:6] + t.actor_buffer[:6]
    X, Y = t._critic_problem(ids)
    for i, tid in enumerate(ids):
        assert np.abs(Y[i].numpy() - book.critic_label(tid, net_eval, net_eval)).max() < 1e-12
    newq = t._new_q(t._rows(ids), t._idx(ids)).numpy()
    assert np.allclose(newq, [book.new_q(tid, net_eval) for tid in ids], rtol=0, atol=1e-12)
    fails = [tid for tid in ids if book.flags[tid] & 1]
    assert fails and all(abs(book.new_q(tid, net_eval) - float(book.mem[tid][0]) * (1 - t.discount)) < 1e-15 for tid in fails)
    # actor: candidate filter and labels
    t.stage_train = True; t.actor_batch_buffer = []; book.actor_batch = []
    st = t.rng.get_state()
    t.UpdateActorBatchBuffer()
    t.rng.set_state(st)
    drawn = []
    n = len(t.actor_buffer)
    for _ in range(min(t.batch, n)):
        c = t.actor_buffer[int(t.rng.randint(0, n))]
        if c not in drawn: drawn.append(c)
    assert t.actor_batch_buffer == [c for c in drawn if book.actor_accepts(c, net_eval)]
    off, sc = book.offset_scale()
    t.UpdateOffsetScale()
    io, isc, _, _ = t.GetOffsetScale()
    assert np.allclose(io, off, atol=1e-12) and np.allclose(isc, sc, rtol=1e-10)


def test_solver_step_is_the_caffe_sgd_rule():
    from oracle import trainer_ref as ref
    t = make_trainer()
    rng = np.random.RandomState(5)
    rows, flags = random_rows(rng, 40)
    t.AddTuples(rows, flags)
    t.SetInputOffsetScale(rng.normal(0, 0.1, S), rng.uniform(0.5, 2, S)); t.SetOutputOffsetScale(rng.normal(0, 0.1, 90), rng.uniform(0.5, 2, 90))
    ids = list(range(32))
    X, Y = t._critic_problem(ids)
    io, isc, oo, osc = [torch.as_tensor(v) for v in t.GetOffsetScale()]
    for step in range(2):                                          # second step exercises the momentum history
        w0 = [b.detach().clone() for b in t.net.blobs()]; h0 = [h.clone() for h in t.history]
        out = t.net((X.double() + io) * isc)
        label = (Y + oo) * osc
        loss = 0.5 * ((out - label) ** 2).sum() / 32
        grads = torch.autograd.grad(loss, t.net.blobs())
        got = t._solver_step(X, Y)
        assert abs(got - float(loss)) < 1e-12
        for b, w, g, h, hn, (lm, dm) in zip(t.net.blobs(), w0, grads, h0, t.history, t.net.blob_mults):
            w_ref, h_ref = ref.caffe_sgd_step(w.numpy(), g.numpy(), h.numpy(), 0.001, 0.9, 0.0005, lm, dm)
            assert np.abs(b.detach().numpy() - w_ref).max() < 1e-15 and np.abs(hn.numpy() - h_ref).max() < 1e-15
    # loss scaling: central finite difference of a few weights against autograd (EuclideanLoss = 1/(2N) sum ||.||^2)
    blob = t.net.blobs()[8]                                        # ip0 weights
    out = t.net((X.double() + io) * isc); label = (Y + oo) * osc
    g = torch.autograd.grad(0.5 * ((out - label) ** 2).sum() / 32, blob)[0]
    with torch.no_grad():
        for idx in [(0, 0), (5, 70), (200, 146)]:
            old = float(blob[idx]); eps = 1e-6
            blob[idx] = old + eps; lp = float(0.5 * ((t.net((X.double() + io) * isc) - label) ** 2).sum() / 32)
            blob[idx] = old - eps; lm_ = float(0.5 * ((t.net((X.double() + io) * isc) - label) ** 2).sum() / 32)
            blob[idx] = old
            assert abs((lp - lm_) / (2 * eps) - float(g[idx])) < 1e-6 * max(1.0, abs(float(g[idx])))


def make_ref_trainer(om, t, seed, **kw):
    """The numpy fp64 restatement (oracle/trainer_ref.py) configured like product trainer `t`: topology from the oracle's own prototxt reader, per-blob
    lr / decay multipliers and solver constants as the train / solver prototxts state them."""
    from oracle import trainer_ref as ref
    d = om.parse_deploy_prototxt(DEPLOY)
    net = ref.RefMaceNet(d.n_terrain, d.n_char, [(d.conv_ch[i], d.conv_k[i]) for i in range(3)], d.fc_terr, d.fc_trunk, d.fc_head, d.n_frags, d.frag_size)
    assert net.num_params == t.net.num_params()
    mults = [(1.0, 1.0), (2.0, 1.0)] * 3 + [(1.0, 1.0), (2.0, 0.0)] * 10          # dog_mace3_train.prototxt: conv blocks carry lr_mult only, ip blocks lr 1/2, decay 1/0
    assert mults == [tuple(m) for m in t.net.blob_mults]
    solver = dict(base_lr=0.001, momentum=0.9, weight_decay=0.0005)               # dog_mace3_solver.prototxt
    return ref.RefMaceTrainer(net, mults, S, A, t.mem_size, 32, t.discount, t.num_init_samples, solver, seed, **kw)


def test_numpy_net_forward_backward_vs_torch_autograd(om):
    """The restatement's hand-derived backward pass against autograd on the product net (fp64, same weights)."""
    t = make_trainer()
    r = make_ref_trainer(om, t, 0)
    w = t.net.flat.detach().numpy().astype(np.float64)
    rng = np.random.RandomState(4)
    x = rng.normal(0, 1, (5, S)); dy = rng.normal(0, 1, (5, 90))
    y = r.net.forward(w, x, keep=True)
    xt = torch.as_tensor(x)
    yt = t.net(xt)
    assert np.abs(y - yt.detach().numpy()).max() < 1e-12
    g = torch.autograd.grad((yt * torch.as_tensor(dy)).sum(), t.net.blobs())
    g_ref = r.net.backward(dy)
    g_t = np.concatenate([v.numpy().reshape(-1) for v in g])
    assert np.abs(g_ref - g_t).max() < 1e-11 * max(1.0, np.abs(g_t).max())


def test_trainer_iterations_match_the_numpy_restatement(om):
    """Six Train() calls of the product trainer (CPU, fp64) vs the whole-trainer numpy restatement: same minibatches (same index stream), critic targets,
    actor candidate filter, labels, normaliser, SGD with momentum / decay -> same weights."""
    rng = np.random.RandomState(9)
    rows, flags = random_rows(rng, 200, p_actor=0.5)
    t = make_trainer(mem_size=256, num_init_samples=100, seed=21)
    r = make_ref_trainer(om, t, 21)
    w0 = t.GetWeights()
    r.set_weights(w0); t.SetWeights(w0)
    t.AddTuples(rows, flags); r.add_tuples(rows, flags)
    for k in range(6):
        t.Train(); r.train()
        assert (t.GetIter(), t.actor_iter) == (r.iter, r.actor_iter) and t.actor_batch_buffer == r.book.actor_batch, k
        assert abs(t.last_loss - r.last_loss) < 1e-9 * max(1.0, abs(r.last_loss))
    a = t.net.flat.detach().numpy()
    assert r.iter == 6 and r.actor_iter >= 1
    assert np.abs(a - r.w).max() < 1e-10 * np.abs(r.w).max() and np.abs(a - w0).max() > 1e-4
    io, isc, _, _ = t.GetOffsetScale()
    assert np.allclose(io, r.in_off, atol=1e-12) and np.allclose(isc, r.in_scale, rtol=1e-10)


def test_stages_iterations_and_target_freeze():
    t = make_trainer(mem_size=256, num_init_samples=64, freeze_target_iters=3, dtype=torch.float32)
    rng = np.random.RandomState(2)
    rows, flags = random_rows(rng, 63, p_actor=0.5)
    t.AddTuples(rows, flags); t.Train()
    assert not t.stage_train and t.GetIter() == 0                  # below trainer_num_init_samples: nothing happens
    rows, flags = random_rows(rng, 150, p_actor=0.5)
    t.AddTuples(rows, flags)
    w_before = t.GetWeights().copy()
    t.Train()
    io, isc, _, _ = t.GetOffsetScale()
    assert t.stage_train and t.GetIter() == 1 and not np.allclose(io, 0) and not np.array_equal(t.GetWeights(), w_before)
    tgt0 = t.target.get_flat().copy()
    losses = []
    for _ in range(8):
        t.Train(); losses.append(t.last_loss)
    assert t.GetIter() == 9 and np.all(np.isfinite(losses))
    assert not np.array_equal(t.target.get_flat(), tgt0)           # refreshed at iterations 3 and 6 ...
    assert not np.array_equal(t.target.get_flat(), t.GetWeights()) # ... and frozen in between
    assert t.actor_iter >= 1                                       # advantage-filtered actor batches were trained
    from deepterrainrl_amd import trainer as tr
    assert tr.anneal(0, 100, 0.9, 0.2) == 0.9 and tr.anneal(250, 100, 0.9, 0.2) == 0.2 and abs(tr.anneal(50, 100, 0.9, 0.2) - 0.55) < 1e-15


def test_train_loop_end_to_end_on_cpu(da):
    """cScenarioTrain loop wiring (rollout -> drain -> AddTuples/Train -> SetPolicy/SetExplore) on the lane-loop test backend."""
    from deepterrainrl_amd import train_loop
    from conftest import EmulScenario
    a = train_loop.parse_arg_file(os.path.join(REFDATA, "args/opt_args_train_mace.txt"))
    assert a["trainer_replay_mem_size"] == "500000" and a["tuple_buffer_size"] == "32" and a["init_exp_temp"] == "20"
    st = train_loop.train("args/opt_args_train_mace.txt", REFDATA, num_envs=64, max_frames=75, trainer_device="cpu", scenario_cls=EmulScenario,
                          extra_args={"terrain_seed": 3, "trainer_num_init_samples": 30, "trainer_replay_mem_size": 512, "trainer_freeze_target_iters": 4, "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"})   # (with the file's 0.9 / 0.9 nearly every early tuple is an actor-exploration tuple and the critic buffer fills slowly, as in the reference)
    assert st["frames"] == 75 and st["tuples"] >= 30 and st["iters"] >= 1
    assert np.all(np.isfinite(st["weights"])) and st["weights"].size == 570474
    io, isc, oo, osc = st["offset_scale"]
    assert np.allclose(io, 0) and np.all(isc == 1)                     # identity input normaliser here: estimated from 30 tuples a near-constant terrain feature gets a scale of 1e9 and the float32 net overflows on its first batch (cNeuralNet::CalcOffsetScale has no floor either); UpdateOffsetScale is covered by the unit tests above
    # overlapped schedule (dtrl_step_begin / dtrl_step_end): same data path, policy one frame staler
    st2 = train_loop.train("args/opt_args_train_mace.txt", REFDATA, num_envs=64, max_frames=75, trainer_device="cpu", scenario_cls=EmulScenario, overlap=True,
                           extra_args={"terrain_seed": 3, "trainer_num_init_samples": 30, "trainer_replay_mem_size": 512, "trainer_freeze_target_iters": 4, "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"})
    assert st2["frames"] == 75 and st2["tuples"] >= 30 and st2["iters"] >= 1 and np.all(np.isfinite(st2["weights"]))


@pytest.mark.parametrize("arg,nparams", [("args/opt_args_train_goat_mace.txt", 570474), ("args/opt_args_train_raptor_mace.txt", 568039)])
def test_train_loop_other_characters(da, arg, nparams):
    """The goat (dog nets, 1 substep per env-step, cliffs) and raptor (275-state / 28-parameter fragments) training configurations."""
    from deepterrainrl_amd import train_loop
    from conftest import EmulScenario
    st = train_loop.train(arg, REFDATA, num_envs=48, max_frames=70, trainer_device="cpu", scenario_cls=EmulScenario,
                          extra_args={"terrain_seed": 2, "trainer_num_init_samples": 30, "trainer_replay_mem_size": 512, "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"})
    assert st["frames"] == 70 and st["tuples"] >= 40 and st["iters"] >= 1 and st["weights"].size == nparams and np.all(np.isfinite(st["weights"]))


@pytest.mark.gpu
def test_gpu_trainer_matches_the_numpy_restatement(om):
    """The trainer as it runs in production (cuda, fp32, HIP-graph replay of evaluation and solver step) against oracle/trainer_ref.py (numpy fp64, its
    own forward / backward / solver): six iterations from the same weights and tuples. Tolerance 2e-4 of the largest weight: fp32 arithmetic over
    six momentum steps (the fp64 CPU twin of this test, test_trainer_iterations_match_the_numpy_restatement, holds 1e-10)."""
    rng = np.random.RandomState(9)
    rows, flags = random_rows(rng, 200, p_actor=0.5)
    t = make_trainer(mem_size=256, num_init_samples=100, device="cuda", dtype=torch.float32, seed=21)
    r = make_ref_trainer(om, t, 21)
    w0 = t.GetWeights()
    r.set_weights(w0); t.SetWeights(w0)
    t.AddTuples(rows, flags); r.add_tuples(rows, flags)
    for k in range(6):
        t.Train(); r.train()
        assert (t.GetIter(), t.actor_iter) == (r.iter, r.actor_iter), k          # same candidate decisions (none sits within fp32 noise of its threshold here)
    a = t.GetWeights().astype(np.float64)
    assert r.iter == 6 and r.actor_iter >= 1
    assert np.abs(a - r.w).max() < 2e-4 * np.abs(r.w).max() and np.abs(a - w0).max() > 1e-4
    assert abs(t.last_loss - r.last_loss) < 1e-3 * max(1.0, abs(r.last_loss))
    assert t.mem.is_cuda and t.net.mods[0].weight.is_cuda


@pytest.mark.gpu
def test_gpu_rollout_to_trainer_loop(da, om):
    """End to end on the GPU: exploration rollouts -> drained MACE rows -> trainer -> weights back into the rollout engine."""
    from deepterrainrl_amd import trainer as tr
    pol = dog_policy(om)
    b = da.BatchScenario("args/opt_args_train_mace.txt", 256, data_root=REFDATA, extra_args={"terrain_seed": 5})
    t = tr.MACETrainer(TRAIN, SOLVER, b.S, b.A, mem_size=4096, num_init_samples=200, freeze_target_iters=50, seed=1)
    off, sc = b.BuildNNOutputOffsetScale()
    t.SetOutputOffsetScale(off, sc)                                 # cScenarioTrain::SetupTrainerOutputOffsetScale
    b.SetExplore(1, 0.3, 20.0, 0.1)                                 # (the file's init 0.9 / 0.9 makes nearly every early tuple an actor tuple)
    b.SetPolicy(t.GetWeights(), *t.GetOffsetScale())
    for f in range(60):
        b.Update()
        rows, flags, ids = b.DrainTuples()
        if len(rows):
            t.AddTuples(rows, flags)
            for _ in range(len(rows) // 32 + 1):
                t.Train()
            b.SetPolicy(t.GetWeights(), *t.GetOffsetScale())
    assert t.GetNumTuples() > 200 and t.GetIter() > 5 and np.isfinite(t.last_loss)
    assert np.all(np.isfinite(t.GetWeights())) and np.all(np.isfinite(b.PoseVel()[0]))


# ---- the Q head's trainer (cQNetTrainer) --------------------------------------------------------------------------------------------
QTRAIN, QSOLVER, QDEPLOY = (os.path.join(NETS, "dog_q_%s.prototxt" % k) for k in ("train", "solver", "deploy"))
QA = 8


def make_q_trainer(**kw):
    from deepterrainrl_amd import trainer as tr
    args = dict(mem_size=256, num_init_samples=100, device="cpu", dtype=torch.float64, seed=5)
    args.update(kw)
    return tr.QNetTrainer(QTRAIN, QSOLVER, S, QA, **args)


def make_ref_q_trainer(om, t, seed):
    from oracle import trainer_ref as ref
    d = om.parse_deploy_prototxt(QDEPLOY)
    net = ref.RefQNet(d.n_terrain, d.n_char, [(d.conv_ch[i], d.conv_k[i]) for i in range(3)], d.fc_terr, d.fc_trunk, d.fc_head, d.frag_size)
    assert net.num_params == t.net.num_params()
    mults = [(1.0, 1.0), (2.0, 1.0)] * 3 + [(1.0, 1.0), (2.0, 0.0)] * 4           # dog_q_train.prototxt
    assert mults == [tuple(m) for m in t.net.blob_mults]
    return ref.RefQTrainer(net, mults, S, QA, t.mem_size, 32, t.discount, t.num_init_samples, dict(base_lr=0.001, momentum=0.9, weight_decay=0.0005), seed)


def q_rows(rng, n, p_fail=0.25):
    rows = rng.normal(0, 1, size=(n, 1 + 2 * S + QA)).astype(np.float32)
    rows[:, 0] = rng.uniform(0, 1, n)
    rows[:, 1 + S:1 + S + QA] = np.eye(QA, dtype=np.float32)[rng.randint(0, QA, n)]     # cBaseControllerQ::RecordPoliAction: one-hot
    return rows, (rng.uniform(size=n) < p_fail).astype(np.int64)


def test_q_net_matches_the_engine_weight_layout_and_numpy_backward(om, da):
    """The trainer's Q net takes the same flat blob vector the rollout engine takes (dtrl_set_policy for -char_ctrl= dog + dog_q_deploy.prototxt), and the
    restatement's forward / backward agree with torch autograd."""
    t = make_q_trainer()
    r = make_ref_q_trainer(om, t, 0)
    from conftest import EmulScenario
    b = EmulScenario("args/opt_args_train_q.txt", 1, data_root=REFDATA, extra_args={"terrain_seed": 1})
    assert b.PolicyNumParams() == t.net.num_params() == r.net.num_params and b.A == QA
    w = t.net.flat.detach().numpy().astype(np.float64)
    rng = np.random.RandomState(3)
    x = rng.normal(0, 1, (4, S)); dy = rng.normal(0, 1, (4, QA))
    y = r.net.forward(w, x, keep=True)
    yt = t.net(torch.as_tensor(x))
    assert np.abs(y - yt.detach().numpy()).max() < 1e-12
    g = torch.autograd.grad((yt * torch.as_tensor(dy)).sum(), t.net.blobs())
    g_t = np.concatenate([v.numpy().reshape(-1) for v in g])
    assert np.abs(r.net.backward(dy) - g_t).max() < 1e-11 * max(1.0, np.abs(g_t).max())


def test_q_trainer_iterations_match_the_numpy_restatement(om):
    rng = np.random.RandomState(12)
    rows, flags = q_rows(rng, 300)                                       # wraps the 256-slot ring
    rows[7, 3] = np.nan                                                  # CheckTuple rejects it
    t = make_q_trainer(seed=8)
    r = make_ref_q_trainer(om, t, 8)
    w0 = t.GetWeights(); t.SetWeights(w0); r.w = w0.astype(np.float64)
    t.AddTuples(rows[:90], flags[:90]); r.add_tuples(rows[:90], flags[:90])
    t.Train(); r.train()
    assert t.GetIter() == r.iter == 0 and not t.stage_train                                  # still collecting initial samples
    t.AddTuples(rows[90:], flags[90:]); r.add_tuples(rows[90:], flags[90:])
    assert (t.head, t.num_tuples) == (r.head, r.num) and np.array_equal(t.mem.numpy(), r.mem)
    for k in range(6):
        t.Train(); r.train()
        assert t.GetIter() == r.iter == k + 1
        assert abs(t.last_loss - r.last_loss) < 1e-9 * max(1.0, abs(r.last_loss)), k
    a = t.net.flat.detach().numpy()
    assert np.abs(a - r.w).max() < 1e-10 * np.abs(r.w).max() and np.abs(a - w0).max() > 1e-4
    io, isc, _, _ = t.GetOffsetScale()
    assert np.allclose(io, r.in_off, atol=1e-12) and np.allclose(isc, r.in_scale, rtol=1e-10)
    # more rows than slots in ONE call: every slot ends up with the LAST row written to it, as if the rows had arrived one by one
    t2 = make_q_trainer(seed=8)
    t2.AddTuples(rows, flags)
    ok = np.all(np.isfinite(rows), axis=1)
    r2 = make_ref_q_trainer(om, t2, 8); r2.add_tuples(rows, flags)
    assert (t2.head, t2.num_tuples) == (r2.head, r2.num) and np.array_equal(t2.mem.numpy(), r2.mem) and np.array_equal(t2.flags, r2.flags) and ok.sum() == 299


@pytest.mark.gpu
def test_gpu_q_trainer_matches_the_numpy_restatement(om):
    rng = np.random.RandomState(12)
    rows, flags = q_rows(rng, 300)
    t = make_q_trainer(seed=8, device="cuda", dtype=torch.float32)
    r = make_ref_q_trainer(om, t, 8)
    w0 = t.GetWeights(); t.SetWeights(w0); r.w = w0.astype(np.float64)
    t.AddTuples(rows, flags); r.add_tuples(rows, flags)
    for k in range(6):
        t.Train(); r.train()
    a = t.GetWeights().astype(np.float64)
    assert t.GetIter() == r.iter == 6
    assert np.abs(a - r.w).max() < 2e-4 * np.abs(r.w).max() and np.abs(a - w0).max() > 1e-4
    assert abs(t.last_loss - r.last_loss) < 1e-3 * max(1.0, abs(r.last_loss))


def test_q_train_loop_end_to_end_on_cpu(da):
    """args/opt_args_train_q.txt through the whole loop: Q-head rollouts (one-hot tuple actions) -> QNetTrainer -> weights and exploration rate back.
    (The input normaliser is left at identity here: estimated from a few dozen tuples instead of the file's 50 000 it scales near-constant terrain
    features by 1e3 and the bootstrapped targets diverge within 20 iterations -- the file's own setting is exercised by the unit tests above.)"""
    from deepterrainrl_amd import train_loop
    from conftest import EmulScenario
    st = train_loop.train("args/opt_args_train_q.txt", REFDATA, num_envs=48, max_frames=100, trainer_device="cpu", scenario_cls=EmulScenario,
                          extra_args={"terrain_seed": 3, "trainer_num_init_samples": 60, "trainer_replay_mem_size": 512, "trainer_init_input_offset_scale": "false"})
    assert st["frames"] == 100 and st["tuples"] >= 60 and st["iters"] >= 10
    assert np.all(np.isfinite(st["weights"])) and st["weights"].size == 461208
    io, isc, oo, osc = st["offset_scale"]
    assert len(oo) == 8 and np.all(oo == -0.5) and np.all(osc == 2)


# ---- the CACLA trainer (cCaclaTrainer / cACTrainer) ---------------------------------------------------------------------------------------
CRITIC = [os.path.join(NETS, "dog_critic_%s.prototxt" % k) for k in ("train", "solver", "deploy")]
ACTOR = [os.path.join(NETS, "dog_actor_%s.prototxt" % k) for k in ("train", "solver", "deploy")]
CA = 29


def make_cacla_trainer(**kw):
    from deepterrainrl_amd import trainer as tr
    args = dict(mem_size=256, num_init_samples=100, freeze_target_iters=3, device="cpu", dtype=torch.float64, seed=4)
    args.update(kw)
    return tr.CaclaTrainer(CRITIC[0], CRITIC[1], ACTOR[0], ACTOR[1], S, CA, **args)


def make_ref_cacla_trainer(om, t, seed):
    from oracle import trainer_ref as ref
    nets = []
    for dep in (CRITIC[2], ACTOR[2]):
        d = om.parse_deploy_prototxt(dep)
        nets.append(ref.RefQNet(d.n_terrain, d.n_char, [(d.conv_ch[i], d.conv_k[i]) for i in range(3)], d.fc_terr, d.fc_trunk, d.fc_head, d.frag_size))
    mults = [(1.0, 1.0), (2.0, 1.0)] * 3 + [(1.0, 1.0), (2.0, 0.0)] * 4
    assert mults == [tuple(m) for m in t.net.blob_mults] == [tuple(m) for m in t.actor.net.blob_mults]
    assert nets[0].num_params == t.net.num_params() and nets[1].num_params == t.actor.net.num_params()
    return ref.RefCaclaTrainer(nets[0], mults, nets[1], mults, S, CA, t.mem_size, 32, 32, t.discount, t.num_init_samples,
                               dict(base_lr=0.001, momentum=0.9, weight_decay=0.0005), seed, freeze_target_iters=t.freeze_target_iters)


def cacla_rows(rng, n, p_off=0.5, p_fail=0.2):
    rows = rng.normal(0, 1, size=(n, 1 + 2 * S + CA)).astype(np.float32)
    rows[:, 0] = rng.uniform(0, 1, n)
    flags = (rng.uniform(size=n) < p_off) * 2 + (rng.uniform(size=n) < p_fail) * 1
    return rows, flags.astype(np.int64)


def run_cacla_trainer_vs_restatement(om, device, dtype, tol, iters=14, factory=None):
    rng = np.random.RandomState(21)
    rows, flags = cacla_rows(rng, 330)                                   # wraps the 256-slot ring: buffers are purged of overwritten slots
    t = factory(device=device, seed=13) if factory else make_cacla_trainer(device=device, dtype=dtype, seed=13)
    r = make_ref_cacla_trainer(om, t, 13)
    wc0, wa0 = t.GetCriticWeights(), t.GetWeights()
    t.SetCriticWeights(wc0); t.SetWeights(wa0)
    r.wc = wc0.astype(np.float64); r.wc_target = r.wc.copy(); r.wa = wa0.astype(np.float64)
    oo = rng.normal(0, 0.1, CA); osc = rng.uniform(0.5, 2, CA)
    t.SetOutputOffsetScale(oo, osc); r.a_out_off, r.a_out_scale = oo, osc
    slots = t.AddTuples(rows[:200], flags[:200]); assert list(slots) == r.add_tuples(rows[:200], flags[:200])
    assert t.off_policy_buffer == r.off_policy
    for k in range(iters):
        t.Train(); r.train()
        if k == 3:                                                      # new tuples arrive mid-training and overwrite old slots
            slots = t.AddTuples(rows[200:], flags[200:]); assert list(slots) == r.add_tuples(rows[200:], flags[200:])
            assert t.off_policy_buffer == r.off_policy
        assert (t.GetIter(), t.actor_iter) == (r.iter, r.actor_iter), k
        if dtype == torch.float64:
            assert t.actor_batch_buffer == r.actor_buf and np.allclose(t.actor_batch_td, r.actor_td, rtol=0, atol=1e-9), k
    assert r.iter == iters and r.actor_iter >= 2, r.actor_iter
    wc, wa = t.GetCriticWeights().astype(np.float64), t.GetWeights().astype(np.float64)
    if dtype == torch.float64:
        wc, wa = t.net.flat.detach().cpu().numpy(), t.actor.net.flat.detach().cpu().numpy()
    assert np.abs(wc - r.wc).max() < tol * np.abs(r.wc).max() and np.abs(wa - r.wa).max() < tol * np.abs(r.wa).max()
    assert np.abs(wc - wc0).max() > 1e-4 and np.abs(wa - wa0).max() > 1e-4
    io, isc, oo2, osc2 = t.GetOffsetScale()
    assert np.allclose(io, r.in_off, atol=1e-6) and np.allclose(oo2, oo, atol=1e-6)
    cio, _, coo, cosc = t.GetCriticOffsetScale()
    assert np.allclose(cio, r.in_off, atol=1e-6) and coo[0] == -0.5 and cosc[0] == 2


def test_cacla_trainer_iterations_match_the_numpy_restatement(om):
    run_cacla_trainer_vs_restatement(om, "cpu", torch.float64, 1e-10)


@pytest.mark.gpu
def test_gpu_cacla_trainer_matches_the_numpy_restatement(om):
    run_cacla_trainer_vs_restatement(om, "cuda", torch.float32, 3e-4)


def test_cacla_train_loop_end_to_end_on_cpu(da):
    """args/opt_args_train_cacla.txt through the whole loop: CACLA rollouts (actor on the device, off-policy flags) -> critic + actor updates -> the
    actor's weights and normalisers back into the engine. (Identity input normaliser for the same reason as in the Q loop test.)"""
    from deepterrainrl_amd import train_loop
    from conftest import EmulScenario
    st = train_loop.train("args/opt_args_train_cacla.txt", REFDATA, num_envs=48, max_frames=100, trainer_device="cpu", scenario_cls=EmulScenario,
                          extra_args={"terrain_seed": 3, "trainer_num_init_samples": 60, "trainer_replay_mem_size": 512, "trainer_init_input_offset_scale": "false"})
    assert st["frames"] == 100 and st["tuples"] >= 60 and st["iters"] >= 10
    assert np.all(np.isfinite(st["weights"])) and st["weights"].size == 463917           # the actor's blobs: what dtrl_set_policy takes for dog_cacla
    io, isc, oo, osc = st["offset_scale"]
    assert len(oo) == 29 and np.all(np.isfinite(osc))
