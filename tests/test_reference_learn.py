"""The reference's OWN trainers (cMACETrainer, cQNetTrainer, cCaclaTrainer: /root/reference/learning compiled unchanged into oracle/_ref/libref_learn.so, with the network
behind cNeuralNet supplied by the harness of oracle/reflearn.py) in lock-step with (i) the numpy restatement oracle/trainer_ref.py and (ii) the PRODUCT's trainers
(deepterrainrl_amd.trainer on CPU fp64, deepterrainrl_amd.hip_trainer on the plain-loop check build of the native step) on the same tuple stream and the same random
stream (independent cRand instances with one seed): replay slots and rows, critic / actor / actor-batch buffers after every AddTuple batch and every Train(), stage and
iteration counters, the input normaliser of the stage switch, every minibatch and label the reference hands its solver, the weights after N iterations.

CPU only (needs /root/reference at BUILD time only: the library travels prebuilt). The -m gpu twin runs the HIP trainer against a trace frozen from these runs."""
import os

import numpy as np
import pytest
import torch

from conftest import REPO
import test_trainer as TT

S, A, NF, FS = TT.S, TT.A, TT.NF, TT.FS


@pytest.fixture(scope="module")
def rl():
    from oracle import reflearn
    if not reflearn.available():
        pytest.skip("oracle/_ref/libref_learn.so not built (make -C oracle/_ref_build)")
    return reflearn


def mace_harness(rl, om):
    from oracle import trainer_ref as ref
    d = om.parse_deploy_prototxt(TT.DEPLOY)
    mults = [(1.0, 1.0), (2.0, 1.0)] * 3 + [(1.0, 1.0), (2.0, 0.0)] * 10
    solver = dict(base_lr=0.001, momentum=0.9, weight_decay=0.0005)

    def make(net_file):
        net = ref.RefMaceNet(d.n_terrain, d.n_char, [(d.conv_ch[i], d.conv_k[i]) for i in range(3)], d.fc_terr, d.fc_trunk, d.fc_head, d.n_frags, d.frag_size)
        h = rl.HarnessNet(net, mults, solver, 32)
        h.out_size = d.n_frags * (1 + d.frag_size)
        return h
    return rl.Harness(make)


def book_state(r):
    return dict(critic=r.buffer(0), actor=r.buffer(1), actor_batch=r.buffer(2), head=r.head, num=r.num_stored, iter=r.iter, actor_iter=r.actor_iter, stage=r.stage_train)


@pytest.mark.parametrize("freeze", [0, 2])
def test_reference_mace_trainer_vs_numpy_restatement(rl, om, freeze):
    """cMACETrainer compiled from the reference vs oracle/trainer_ref.RefMaceTrainer: every bookkeeping decision and -- since both sides run the same numpy net --
    the weights to rounding. Tuples arrive in three batches (init stage, the stage switch with UpdateOffsetScale, mid-training arrivals that overwrite ring slots)."""
    rng = np.random.RandomState(9)
    rows, flags = TT.random_rows(rng, 420, p_actor=0.5)
    H = mace_harness(rl, om)
    R = rl.RefTrainer("mace", H, TT.DEPLOY, TT.SOLVER, mem_size=256, num_init_samples=100, discount=0.9, freeze_target_iters=freeze, num_frags=NF, frag_size=FS, seed=77)
    t = TT.make_trainer(mem_size=256, num_init_samples=100, seed=21, freeze_target_iters=freeze)
    r = TT.make_ref_trainer(om, t, 21, freeze_target_iters=freeze)
    r.rng = rl.RefRandStream(77)
    w0 = t.GetWeights().astype(np.float64)
    r.set_weights(w0)
    for i in range(R.num_pool()):
        R.pool_net(i).w = w0.copy()
    assert (R.S, R.A, R.W, R.batch) == (S, A, 1 + 2 * S + A, 32)

    def same(tag):
        b = r.book
        st = book_state(R)
        assert st["critic"] == list(b.critic) and st["actor"] == list(b.actor) and st["actor_batch"] == list(b.actor_batch), tag
        assert (st["head"], st["num"], st["iter"], st["actor_iter"], st["stage"]) == (b.head, b.num, r.iter, r.actor_iter, r.stage_train), tag

    k = 0
    for n_new, n_train in ((60, 2), (150, 5), (210, 6)):
        slots_ref = R.add_rows(rows[k:k + n_new], flags[k:k + n_new])
        slots = r.add_tuples(rows[k:k + n_new], flags[k:k + n_new])
        assert slots_ref == slots
        k += n_new
        same("after add %d" % k)
        for j in range(n_train):
            R.train(); r.train()
            same("train %d/%d" % (k, j))
            assert np.abs(R.pool_net(0).w - r.w).max() <= 1e-12 * np.abs(r.w).max(), (k, j)
    assert r.iter == 11 and r.actor_iter >= 2 and np.abs(r.w - w0).max() > 1e-4
    for tt in (0, 17, 255):
        row, fl = R.mem_row(tt)
        assert np.array_equal(row, r.book.mem[tt]) and fl == r.book.flags[tt]
    io, isc = R.input_offset_scale()
    assert np.allclose(io, r.in_off, rtol=0, atol=1e-13) and np.allclose(isc, r.in_scale, rtol=1e-12)
    if freeze:
        assert R.num_pool() == 2 and np.abs(R.pool_net(1).w - r.w_target).max() <= 1e-12 * np.abs(r.w).max()
    R.close()


def test_reference_learner_route_vs_the_training_loops_chunk_protocol(rl, om):
    """The ENV side of the reference's trainer seam, compiled unchanged: cNeuralNetLearner::Train(tuples) (learning/NeuralNetLearner.cpp:33-46 -- what cScenarioExp calls
    when its tuple buffer is full: AddTuples, Train, SyncNet under the trainer's lock) through a learner the trainer itself handed out (RequestLearner), fed 32 tuples at a
    time. Against it: the restatement driven chunk by chunk, and the PRODUCT's loop protocol train_loop.feed_chunks (AddTuples + Train per -tuple_buffer_size= tuples) on the
    torch trainer. Bookkeeping identical after every chunk; the net the learner synchronised (the controller's, i.e. what the env threads would run) equals the trainer's."""
    from deepterrainrl_amd import train_loop
    rng = np.random.RandomState(9)
    rows, flags = TT.random_rows(rng, 320, p_actor=0.5)
    H = mace_harness(rl, om)
    R = rl.RefTrainer("mace", H, TT.DEPLOY, TT.SOLVER, mem_size=256, num_init_samples=100, discount=0.9, freeze_target_iters=2, num_frags=NF, frag_size=FS, seed=77)
    t = TT.make_trainer(mem_size=256, num_init_samples=100, seed=21, freeze_target_iters=2)
    r = TT.make_ref_trainer(om, t, 21, freeze_target_iters=2)
    r.rng = rl.RefRandStream(77)
    t.rng = rl.RefRandStream(77)
    w0 = t.GetWeights().astype(np.float64)
    t.SetWeights(t.GetWeights())                 # (the fp64 peer starts from the float32-valued blobs as well)
    r.set_weights(w0)
    for i in range(R.num_pool()):
        R.pool_net(i).w = w0.copy()
    for k in range(0, 320, 32):
        learner_net = R.learner_train(rows[k:k + 32], flags[k:k + 32])
        r.add_tuples(rows[k:k + 32], flags[k:k + 32]); r.train()
        train_loop.feed_chunks(t, rows[k:k + 32], flags[k:k + 32], 32)
        st, b = book_state(R), r.book
        assert st["critic"] == list(b.critic) and st["actor"] == list(b.actor) and st["actor_batch"] == list(b.actor_batch), k
        assert (st["head"], st["num"], st["iter"], st["actor_iter"], st["stage"]) == (b.head, b.num, r.iter, r.actor_iter, r.stage_train), k
        assert (R.learner_iter, R.learner_num_tuples) == (r.iter, k + 32) == (t.GetIter(), t.GetNumTuples()), k     # GetNumTuples = mTotalTuples: every tuple ever stored
        assert np.array_equal(learner_net.w, R.pool_net(0).w), k                       # SyncNet: CopyModel of the trainer's net
        assert np.abs(R.pool_net(0).w - r.w).max() <= 1e-12 * np.abs(r.w).max(), k
        assert np.abs(t.net.flat.detach().numpy().astype(np.float64) - r.w).max() <= 1e-10 * np.abs(r.w).max(), k
    assert r.iter == 7 and r.actor_iter >= 1 and np.abs(r.w - w0).max() > 1e-4
    R.close()


def q_harness(rl, om, deploys, out_sizes):
    """single-head nets (dog_q / dog_critic / dog_actor): the prototxt path picks the topology"""
    from oracle import trainer_ref as ref
    mults = [(1.0, 1.0), (2.0, 1.0)] * 3 + [(1.0, 1.0), (2.0, 0.0)] * 4
    solver = dict(base_lr=0.001, momentum=0.9, weight_decay=0.0005)

    def make(net_file):
        key = [k for k in deploys if os.path.basename(deploys[k]) == os.path.basename(net_file)][0]
        d = om.parse_deploy_prototxt(deploys[key])
        net = ref.RefQNet(d.n_terrain, d.n_char, [(d.conv_ch[i], d.conv_k[i]) for i in range(3)], d.fc_terr, d.fc_trunk, d.fc_head, d.frag_size)
        h = rl.HarnessNet(net, mults, solver, 32)
        h.out_size = out_sizes[key]; h.key = key
        return h
    return rl.Harness(make)


def test_reference_q_trainer_vs_numpy_restatement(rl, om):
    """cQNetTrainer compiled from the reference vs RefQTrainer: ring wrap, a row CheckTuple rejects, uniform minibatch draws, Q targets, weights."""
    rng = np.random.RandomState(12)
    rows, flags = TT.q_rows(rng, 300)
    rows[7, 3] = np.nan
    H = q_harness(rl, om, {"q": TT.QDEPLOY}, {"q": TT.QA})
    R = rl.RefTrainer("q", H, TT.QDEPLOY, TT.QSOLVER, mem_size=256, num_init_samples=100, discount=0.9, seed=5)
    t = TT.make_q_trainer(seed=8)
    r = TT.make_ref_q_trainer(om, t, 8)
    r.rng = rl.RefRandStream(5)
    w0 = t.GetWeights().astype(np.float64)
    r.w = w0.copy(); R.pool_net(0).w = w0.copy()
    assert R.add_rows(rows[:90], flags[:90])[7] == -1
    r.add_tuples(rows[:90], flags[:90])
    R.train(); r.train()
    assert R.iter == r.iter == 0 and not R.stage_train
    R.add_rows(rows[90:], flags[90:]); r.add_tuples(rows[90:], flags[90:])
    assert (R.head, R.num_stored) == (r.head, r.num)
    for tt in range(0, 256, 5):
        row, fl = R.mem_row(tt)
        assert np.array_equal(row, r.mem[tt]) and fl == r.flags[tt]
    for k in range(6):
        R.train(); r.train()
        assert R.iter == r.iter == k + 1
        assert np.abs(R.pool_net(0).w - r.w).max() <= 1e-10 * np.abs(r.w).max(), k
    io, isc = R.input_offset_scale()
    assert np.allclose(io, r.in_off, rtol=0, atol=1e-12) and np.allclose(isc, r.in_scale, rtol=1e-10)
    assert np.abs(r.w - w0).max() > 1e-4
    R.close()


def test_reference_cacla_trainer_vs_numpy_restatement(rl, om):
    """cCaclaTrainer (over cACTrainer, eModeCacla) compiled from the reference vs RefCaclaTrainer: off-policy buffer, TD-filtered actor batches with their TD values'
    bookkeeping, critic target refresh, both nets' weights; tuples arrive mid-training and overwrite ring slots."""
    rng = np.random.RandomState(21)
    rows, flags = TT.cacla_rows(rng, 330)
    H = q_harness(rl, om, {"critic": TT.CRITIC[2], "actor": TT.ACTOR[2]}, {"critic": 1, "actor": TT.CA})
    R = rl.RefTrainer("cacla", H, TT.CRITIC[2], TT.CRITIC[1], mem_size=256, num_init_samples=100, discount=0.9, freeze_target_iters=3,
                      actor_net_file=TT.ACTOR[2], actor_solver_file=TT.ACTOR[1], seed=31)
    t = TT.make_cacla_trainer(seed=13)
    r = TT.make_ref_cacla_trainer(om, t, 13)
    r.rng = rl.RefRandStream(31)
    wc0, wa0 = t.GetCriticWeights().astype(np.float64), t.GetWeights().astype(np.float64)
    r.wc = wc0.copy(); r.wc_target = wc0.copy(); r.wa = wa0.copy()
    for i in range(R.num_pool()):
        assert R.pool_net(i).key == "critic"
        R.pool_net(i).w = wc0.copy()
    assert R.actor_net().key == "actor"
    R.actor_net().w = wa0.copy()
    oo = rng.normal(0, 0.1, TT.CA); osc = rng.uniform(0.5, 2, TT.CA)
    R.set_actor_output_offset_scale(oo, osc); r.a_out_off, r.a_out_scale = oo, osc
    R.set_critic_output_offset_scale(np.full(1, -0.5), np.full(1, 2.0))
    assert R.add_rows(rows[:200], flags[:200]) == r.add_tuples(rows[:200], flags[:200])
    assert R.buffer(1) == r.off_policy
    for k in range(14):
        R.train(); r.train()
        if k == 3:
            assert R.add_rows(rows[200:], flags[200:]) == r.add_tuples(rows[200:], flags[200:])
            assert R.buffer(1) == r.off_policy
        assert (R.iter, R.actor_iter) == (r.iter, r.actor_iter) and R.buffer(2) == r.actor_buf, k
        assert np.abs(R.pool_net(0).w - r.wc).max() <= 1e-10 * np.abs(r.wc).max() and np.abs(R.actor_net().w - r.wa).max() <= 1e-10 * np.abs(r.wa).max(), k
    assert r.iter == 14 and r.actor_iter >= 2 and np.abs(r.wa - wa0).max() > 1e-4
    assert np.abs(R.pool_net(1).w - r.wc_target).max() <= 1e-10 * np.abs(r.wc).max()
    R.close()


def run_reference_vs_product_mace(rl, om, t, tol, freeze, get_w=None):
    """The product's MACE trainer against the compiled reference DIRECTLY (no restatement in between): same tuples, the reference's random stream."""
    rng = np.random.RandomState(9)
    rows, flags = TT.random_rows(rng, 420, p_actor=0.5)
    H = mace_harness(rl, om)
    R = rl.RefTrainer("mace", H, TT.DEPLOY, TT.SOLVER, mem_size=256, num_init_samples=100, discount=0.9, freeze_target_iters=freeze, num_frags=NF, frag_size=FS, seed=78)
    t.rng = rl.RefRandStream(78)
    w0 = t.GetWeights().astype(np.float64)
    t.SetWeights(t.GetWeights())
    for i in range(R.num_pool()):
        R.pool_net(i).w = w0.copy()
    k = 0
    for n_new, n_train in ((60, 2), (150, 5), (210, 6)):
        assert R.add_rows(rows[k:k + n_new], flags[k:k + n_new]) == list(t.AddTuples(rows[k:k + n_new], flags[k:k + n_new]))
        k += n_new
        assert (R.buffer(0), R.buffer(1), R.buffer(2)) == (list(t.critic_buffer), list(t.actor_buffer), list(t.actor_batch_buffer)), k
        for j in range(n_train):
            R.train(); t.Train()
            assert (R.iter, R.actor_iter, R.stage_train) == (t.GetIter(), t.actor_iter, t.stage_train), (k, j)
            assert (R.buffer(0), R.buffer(1), R.buffer(2)) == (list(t.critic_buffer), list(t.actor_buffer), list(t.actor_batch_buffer)), (k, j)
    w = get_w(t) if get_w else t.GetWeights().astype(np.float64)
    assert R.iter == 11 and R.actor_iter >= 2
    assert np.abs(w - R.pool_net(0).w).max() < tol * np.abs(w).max() and np.abs(w - w0).max() > 1e-4
    io, isc, _, _ = t.GetOffsetScale()
    rio, risc = R.input_offset_scale()
    assert np.allclose(io, rio, rtol=0, atol=1e-6) and np.allclose(isc, risc, rtol=1e-5)
    R.close()


@pytest.mark.parametrize("freeze", [2])      # (freeze 0 is covered against the numpy restatement above and against the native step below; the torch peer takes 20 s per case)
def test_reference_mace_trainer_vs_product_torch_trainer(rl, om, freeze):
    run_reference_vs_product_mace(rl, om, TT.make_trainer(mem_size=256, num_init_samples=100, seed=21, freeze_target_iters=freeze), 1e-10, freeze,
                                  get_w=lambda t: t.net.flat.detach().numpy().astype(np.float64))


@pytest.mark.parametrize("freeze", [0, 2])
def test_reference_mace_trainer_vs_native_step_check_build(rl, om, freeze):
    """the native step (plain-loop check build of the HIP trainer's operand definitions, float32) against the compiled reference"""
    import test_hip_trainer as TH
    run_reference_vs_product_mace(rl, om, TH.make_native(TH.EMUL_TRAINER_LIB, mem_size=256, num_init_samples=100, seed=21, freeze_target_iters=freeze), 3e-4, freeze)


def run_reference_trainer_on_the_product_nets(rl, om, lib_path, product, tol):
    """INTEGRATION.md 4b made concrete: the reference's OWN cMACETrainer (compiled unchanged) with every cNeuralNet forwarded to include/BatchNeuralNet.h's
    cBatchNeuralNet over the native trainer step (oracle/_ref/libref_learn_native*.so) -- its replay memory, buffers, draws, labels and schedule, the product's
    forward / backward / Caffe SGD -- against hip_trainer.HipMACETrainer, where the product's own host bookkeeping drives the same native step: same tuples, same
    random stream -> same buffers and counters at every Train(), weights within float32 rounding of the label arithmetic (double in the reference's trainer)."""
    from conftest import REFDATA
    L = rl.lib(lib_path)
    L.ref_learn_native_config(REFDATA.encode(), -1)
    rng = np.random.RandomState(9)
    rows, flags = TT.random_rows(rng, 420, p_actor=0.5)
    R = rl.RefTrainer("mace", None, TT.DEPLOY, TT.SOLVER, mem_size=256, num_init_samples=100, discount=0.9, freeze_target_iters=2, num_frags=NF, frag_size=FS, seed=78, lib_path=lib_path)
    assert (R.S, R.A, R.batch, R.num_pool()) == (S, A, 32, 2)
    t = product
    t.rng = rl.RefRandStream(78)
    w0 = t.GetWeights().copy()
    t.SetWeights(w0)
    for i in range(R.num_pool()):
        R.set_pool_weights(i, w0)
    k = 0
    for n_new, n_train in ((60, 2), (150, 5), (210, 6)):
        assert R.add_rows(rows[k:k + n_new], flags[k:k + n_new]) == list(t.AddTuples(rows[k:k + n_new], flags[k:k + n_new]))
        k += n_new
        for j in range(n_train):
            R.train(); t.Train()
            assert (R.iter, R.actor_iter, R.stage_train) == (t.GetIter(), t.actor_iter, t.stage_train), (k, j)
            assert (R.buffer(0), R.buffer(1), R.buffer(2)) == (list(t.critic_buffer), list(t.actor_buffer), list(t.actor_batch_buffer)), (k, j)
    a, b = R.pool_weights(0).astype(np.float64), t.GetWeights().astype(np.float64)
    assert R.iter == 11 and R.actor_iter >= 2
    assert np.abs(a - b).max() < tol * np.abs(b).max() and np.abs(b - w0).max() > 1e-4
    R.close()


def test_reference_mace_trainer_running_on_the_product_nets(rl, om):
    if not os.path.exists(rl.NATIVE_LIB_PATH):
        pytest.skip("oracle/_ref/libref_learn_native.so not built")
    import test_hip_trainer as TH
    run_reference_trainer_on_the_product_nets(rl, om, rl.NATIVE_LIB_PATH, TH.make_native(TH.EMUL_TRAINER_LIB, mem_size=256, num_init_samples=100, seed=21, freeze_target_iters=2), 3e-4)


def test_batch_neural_net_shim_inside_the_reference_tree():
    """include/BatchNeuralNet.h compiled against /root/reference's headers (tests/shim/Makefile) and driven through cNeuralNet's calls on the plain-loop check build"""
    import subprocess
    from conftest import REFDATA
    exe = os.path.join(REPO, "tests", "shim", "drive_shim_net_emul")
    if not os.path.exists(exe):
        pytest.skip("tests/shim/drive_shim_net_emul not built")
    r = subprocess.run([exe, REFDATA, "/tmp"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "shim net ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
