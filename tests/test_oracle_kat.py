"""Oracle pinned against every known-answer the reference ships for the rollout path.
Trajectory goldens do not exist in the reference (Bullet is absent): the dynamics stay 'parity unpinned'."""
import json
import os

import numpy as np
import pytest

from conftest import REFDATA, REFERENCE, GOLDEN, dog_policy


def test_dog_structure(om):
    m, info = om.build_model("args/sim_dog_args.txt", REFDATA)
    assert m.L == 21 and m.D == 23
    assert list(m.parent[:21]) == [-1, 0, 1, 2, 3, 4, 5, 6, 7, 0, 9, 10, 11, 5, 13, 14, 15, 0, 17, 18, 19]
    assert abs(sum(m.body_mass[:21]) - 33.67) < 1e-9
    assert info["S"] == 283                      # dog_mace3_deploy input_dim
    assert m.n_params == 30 and m.n_actions == 8 and m.n_sets == 3 and m.default_action == 0
    free = [j for j in range(21) if m.lim_lo[j] > m.lim_hi[j]]
    assert free == [0, 9, 10, 11, 12]            # root + tail0-3
    assert [j for j in range(21) if m.use_world[j]] == [13, 17]   # shoulder, hip
    assert m.num_update_steps == 20 and m.num_sim_substeps == 5 and m.world_scale == 4


def test_goat_structure(om):
    m, _ = om.build_model("args/opt_args_train_goat_mace.txt", REFDATA)
    assert m.L == 21 and m.D == 23 and m.n_sets == 5 and m.n_actions == 5
    assert m.enable_grav_comp == 0 and m.target_vel_x == 2.0
    assert m.num_sim_substeps == 1 and m.world_scale == 1 and m.valid_init_pos_x == 1 and m.init_pos_x == 1.2


def test_net_topology_and_param_count(om):
    desc = om.parse_deploy_prototxt(os.path.join(REFDATA, "data/policies/dog/nets/dog_mace3_deploy.prototxt"))
    assert (desc.n_terrain, desc.n_char, list(desc.conv_ch), list(desc.conv_k)) == (200, 83, [16, 32, 32], [8, 4, 4])
    assert (desc.fc_terr, desc.fc_trunk, desc.fc_head, desc.n_frags, desc.frag_size) == (64, 256, 128, 3, 29)
    assert om.lib().orc_net_num_params(desc) == 570474


@pytest.mark.parametrize("scale", ["dog_mace3_slopes_mixed_model_scale.txt", "dog_mace3_mixed_model_scale.txt"])
def test_output_offset_scale_matches_shipped_scale_file(om, scale):
    """BuildNNOutputOffsetScale restated from the controller files reproduces the shipped normaliser (printed with 6 decimals)."""
    m, _ = om.build_model("args/dog_slopes_mixed_args.txt", REFDATA)
    _, _, oo, osc = om.load_scale_file(os.path.join(REFDATA, "data/policies/dog/models", scale))
    off, sc = om.build_output_offset_scale(m, 3)
    assert oo.shape == (90,) and np.all(oo[:3] == -0.5) and np.all(osc[:3] == 2)
    assert np.abs(off - oo).max() < 5e-7
    assert np.abs(sc / osc - 1).max() < 2e-4
    assert np.allclose(oo[3:6], [-0.014184, -378.334188, -408.983406])


def test_mass_matrix_matches_kinetic_energy(om):
    """CRBA (restated 6-D algorithm) vs an independent evaluation: 1/2 qd^T H qd == sum 1/2 m |v_c|^2 + 1/2 I w^2."""
    m, _ = om.build_model("args/sim_dog_args.txt", REFDATA)
    e = om.OracleEnv(m, terrain_seed=1)
    rng = np.random.RandomState(0)
    for _ in range(5):
        q = np.array(m.pose0[:23]) + rng.uniform(-0.3, 0.3, 23)
        qd = rng.uniform(-5, 5, 23)
        H, Cq, Ct, g = e.rbd(q, qd)
        assert np.abs(H - H.T).max() < 1e-12 and np.linalg.eigvalsh(H).min() > 0
        e.set_pose_vel(q, qd)
        _, v, _ = e.bodies()
        w = np.array([qd[2] + sum(qd[a + 2] for a in _path(m, j)[1:]) for j in range(21)])
        izz = np.array([m.body_mass[j] / 12.0 * (m.body_size[j][0] ** 2 + m.body_size[j][1] ** 2) for j in range(21)])
        mass = np.array(m.body_mass[:21])
        ke = 0.5 * (mass * (v ** 2).sum(1)).sum() + 0.5 * (izz * w ** 2).sum()
        assert abs(0.5 * qd @ H @ qd - ke) < 1e-9 * max(1.0, ke)


def _path(m, j):
    p = []
    while j >= 0:
        p.append(j); j = m.parent[j]
    return p[::-1]


def test_bias_force_gravity_limit_and_quirk(om):
    """At zero velocity the bias force is minus the generalised gravity force, and the shipped BuildCjPlanar quirk vanishes."""
    m, _ = om.build_model("args/sim_dog_args.txt", REFDATA)
    e = om.OracleEnv(m, terrain_seed=1)
    q = np.array(m.pose0[:23]); z = np.zeros(23)
    H, Cq, Ct, g = e.rbd(q, z)
    assert np.abs(Cq - Ct).max() < 1e-12
    assert np.abs(Ct + g).max() < 1e-9
    assert abs(g[1] + 9.8 * 33.67) < 1e-9 and abs(g[0]) < 1e-12
    qd = np.array(m.vel0[:23])
    _, Cq, Ct, _ = e.rbd(q, qd)
    assert np.abs(Cq - Ct).max() > 1.0     # the quirk is real at speed


def test_terrain_generator_invariants(om):
    m, _ = om.build_model("args/dog_slopes_mixed_args.txt", REFDATA)
    p = np.array(m.terrain_params[0][:40])
    flat = om.terrain_build(0, p, 3, 20.0)
    assert len(flat) == 201 and np.all(flat == 0)              # ceil(20 / 0.1f) + 1 vertices
    gaps = om.terrain_build(1, p, 3, 20.0)
    assert set(np.unique(gaps)) == {np.float32(-2.0), np.float32(0.0)} and gaps[0] == 0 and gaps[-1] == 0
    a = om.terrain_build(11, p, 42, 20.0); b = om.terrain_build(11, p, 42, 20.0); c = om.terrain_build(11, p, 43, 20.0)
    assert np.array_equal(a, b) and not np.array_equal(a[:50], c[:50])
    assert len(a) >= 201


def test_terrain_golden_vectors(om):
    """Regression vectors frozen from the oracle (libstdc++ minstd_rand0 streams): (type, seed) -> heights."""
    g = np.load(os.path.join(GOLDEN, "terrain_golden.npz"))
    m, _ = om.build_model("args/dog_slopes_mixed_args.txt", REFDATA)
    p = np.array(m.terrain_params[0][:40])
    for key in g.files:
        t, s = (int(x) for x in key.split("_")[1:])
        assert np.array_equal(om.terrain_build(t, p, s, 20.0), g[key]), key


def test_golden_trace_sim_dog(om):
    """config[0]: args/sim_dog_args.txt, flat, 1 env, 1200 substeps: the committed q/qd trace is reproduced bit-stably."""
    g = np.load(os.path.join(GOLDEN, "sim_dog_trace.npz"))
    m, _ = om.build_model("args/sim_dog_args.txt", REFDATA)
    e = om.OracleEnv(m, terrain_seed=int(g["terrain_seed"]))
    for k in range(g["q"].shape[0]):
        e.step(20)
        q, qd = e.pose_vel()
        assert np.abs(q - g["q"][k]).max() < 1e-9 and np.abs(qd - g["qd"][k]).max() < 1e-7
    # behavioural plausibility: one bound cycle completes with contact-triggered transitions
    assert e.stats()["cycles"] >= 1 and q[0] > 1.2 and not (e.flags() & 1)


def test_golden_trace_sim_raptor(om):
    g = np.load(os.path.join(GOLDEN, "sim_raptor_trace.npz"))
    m, _ = om.build_model("args/sim_raptor_args.txt", REFDATA)
    e = om.OracleEnv(m, terrain_seed=int(g["terrain_seed"]))
    for k in range(g["q"].shape[0]):
        e.step(20)
        q, qd = e.pose_vel()
        assert np.abs(q - g["q"][k]).max() < 1e-9 and np.abs(qd - g["qd"][k]).max() < 1e-7


def test_raptor_runs_20s_on_flat(om):
    m, _ = om.build_model("args/sim_raptor_args.txt", REFDATA)
    e = om.OracleEnv(m, terrain_seed=1)
    for _ in range(200):
        e.update()
    q, _ = e.pose_vel()
    assert q[0] > 20.0 and not (e.flags() & 1) and e.stats()["cycles"] > 20


def test_dog_runs_30s_on_flat(om):
    m, _ = om.build_model("args/sim_dog_args.txt", REFDATA)
    e = om.OracleEnv(m, terrain_seed=1)
    for _ in range(300):
        e.update()
    q, _ = e.pose_vel()
    assert q[0] > 30.0 and not (e.flags() & 1) and e.stats()["cycles"] > 15


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present (GPU box)")
def test_fixtures_match_reference_inputs(om):
    """tests/golden/refdata is a faithful re-serialisation of the reference's input files for the configs under test."""
    for rel in ["data/characters/dog.txt", "data/characters/goat.txt", "data/states/dog_bound_state.txt", "data/terrain/slopes_mixed.txt",
                "data/controllers/dog/bound.txt", "data/policies/dog/models/dog_mace3_slopes_mixed_model_scale.txt"]:
        ref = json.load(open(os.path.join(REFERENCE, rel))); fix = json.load(open(os.path.join(REFDATA, rel)))
        for k in fix:
            if k == "BodyDefs":
                for a, b in zip(fix[k], ref[k]):
                    assert all(a[f] == b[f] for f in a)
            else:
                assert fix[k] == ref[k], (rel, k)
    for arg in ["args/sim_dog_args.txt", "args/dog_slopes_mixed_args.txt", "args/opt_args_train_mace.txt"]:
        assert om.args_to_dict(om.parse_arg_file(os.path.join(REFERENCE, arg))) == om.args_to_dict(om.parse_arg_file(os.path.join(REFDATA, arg)))
    ma, _ = om.build_model("args/dog_slopes_mixed_args.txt", REFERENCE); mb, _ = om.build_model("args/dog_slopes_mixed_args.txt", REFDATA)
    assert bytes(ma) == bytes(mb)
    da_ = om.parse_deploy_prototxt(os.path.join(REFERENCE, "data/policies/dog/nets/dog_mace3_deploy.prototxt"))
    db_ = om.parse_deploy_prototxt(os.path.join(REFDATA, "data/policies/dog/nets/dog_mace3_deploy.prototxt"))
    assert bytes(da_) == bytes(db_)
