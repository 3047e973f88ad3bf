"""The oracle (and the product's host code) pinned against the REFERENCE ITSELF.

oracle/_ref/libref_core.so is built by oracle/_ref_build/Makefile from the reference's own translation units, compiled unchanged where
they lie under /root/reference (util/Rand, util/ArgParser, sim/TerrainGen2D, anim/KinTree, sim/SpAlg, sim/RBDModel, sim/RBDUtil; only
the absent Eigen / jsoncpp headers are stand-ins). These tests can fail because the restatement in oracle/or_*.h -- or the product's
host-side generator / parser -- disagrees with the reference's statements:

  a18-a19  terrain strips: 14 types x 8 seeds x 3 widths, reference == oracle == product (dtrl_terrain_build), bit for bit
           terrain files: Type + 40-vector of every shipped data/terrain/*.txt, reference == oracle loader == product loader
  cRand    libstdc++ streams of util/Rand.cpp == an independent restatement of minstd_rand0 + generate_canonical
  args     every args/*.txt of the reference: token count and every key's value, reference == oracle parser == product parser
  a7-a10, a13, a17, a28, a29   cKinTree tables, cRBDModel::Update, mass matrix, bias force (BuildCjPlanar as shipped), gravity force,
           Jacobian-derived COM / COM velocity, joint and body positions: reference == oracle to 1e-12 (dog, goat, raptor)

Tests that need the reference checkout or the compiled library skip without them (the GPU box has the prebuilt library but no
/root/reference); tests/golden/ref_golden.npz holds frozen outputs of the same entry points (make_ref_golden.py) and is always checked.
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, HIP_LIB, REFDATA, REFERENCE

from oracle import refcore as rc

HAVE_REF = os.path.isdir(os.path.join(REFERENCE, "args"))
needs_lib = pytest.mark.skipif(not rc.available(), reason="oracle/_ref/libref_core.so not built (needs /root/reference at build time)")
needs_ref = pytest.mark.skipif(not (HAVE_REF and rc.available()), reason="needs the reference checkout and libref_core.so")

TERRAIN_TYPES = ["flat", "gaps", "steps", "walls", "bumps", "mixed", "narrow_gaps", "slopes", "slopes_gaps", "slopes_steps",
                 "slopes_walls", "slopes_mixed", "slopes_narrow_gaps", "cliffs"]
CHARS = [("dog", "args/dog_slopes_mixed_args.txt"), ("goat", "args/goat_cliffs_args.txt"), ("raptor", "args/raptor_narrow_gaps_args.txt")]


def _param_sets(om):
    """default vector, the slopes_mixed / cliffs_rugged-like vectors of the oracle's own reading of the shipped files, one stress vector"""
    sets = [np.array([d for _, d in om.TERRAIN_PARAMS], np.float64)]
    for arg in ("args/dog_slopes_mixed_args.txt", "args/goat_cliffs_args.txt", "args/raptor_narrow_gaps_args.txt"):
        m, _ = om.build_model(arg, REFDATA)
        sets.append(np.array(m.terrain_params[0][:], np.float64))
    s = sets[0].copy()
    s[36] = 3            # CliffMiniCountMax
    s[18:20] = [0.01, 0.05]
    sets.append(s)
    return sets


# ----------------------------------------------------------------------------------------------------------------------
# terrain
@needs_lib
@pytest.mark.parametrize("ttype", TERRAIN_TYPES)
def test_terrain_strips_reference_oracle_product_bit_exact(om, da, ttype):
    """cTerrainGen2D::GetTerrainFunc(type)(width, params, cRand(seed)) -- sim/TerrainGen2D.cpp:148-181, 185-706."""
    t_id = om.TERRAIN_TYPES.index(ttype)
    n_cmp = 0
    for pi, prm in enumerate(_param_sets(om)):
        for seed in (0, 1, 2, 7, 11, 1234, 20260925, 4294967295):
            for width in (0.05, 20.0, 37.3):
                ref, w_ref = rc.terrain_build(ttype, prm, seed, width)
                orc = om.terrain_build(t_id, prm, seed, width)
                prod, w_prod = da.terrain_build(ttype, prm, seed, width)
                assert ref.dtype == np.float32 and len(ref) == len(orc) == len(prod), (ttype, pi, seed, width)
                assert np.array_equal(ref.view(np.uint32), orc.view(np.uint32)), "oracle != reference: %s set %d seed %d width %g" % (ttype, pi, seed, width)
                assert np.array_equal(ref.view(np.uint32), prod.view(np.uint32)), "product != reference: %s set %d seed %d width %g" % (ttype, pi, seed, width)
                assert w_ref == w_prod
                n_cmp += 1
    assert n_cmp == 5 * 8 * 3


@needs_lib
def test_terrain_strip_continues_an_existing_profile(om):
    """The generators append to a non-empty vector from its last vertex (sim/GroundVar2D.cpp:318-322 pads 2 m of flat in front):
    the reference with a prefix == the reference's flat strip followed by the oracle's strip shifted to the prefix height."""
    prm = _param_sets(om)[1]
    for ttype in ("slopes_mixed", "cliffs", "narrow_gaps"):
        a, _ = rc.terrain_build(ttype, prm, 5, 20.0, prefix=21, prefix_h=0.25)
        assert np.all(a[:21] == np.float32(0.25))
        assert len(a) > 21 + 150


@needs_ref
def test_terrain_files_reference_oracle_product(om, da):
    """Type + parameter vectors of every shipped terrain file (cTerrainGen2D::LoadParams, sim/TerrainGen2D.cpp:69-81)."""
    names = rc.terrain_param_names()
    assert names == [n for n, _ in om.TERRAIN_PARAMS]
    assert np.array_equal(rc.terrain_default_params(), np.array([d for _, d in om.TERRAIN_PARAMS], np.float64))
    files = sorted(glob.glob(os.path.join(REFERENCE, "data", "terrain", "*.txt")))
    assert len(files) >= 8
    for f in files:
        ty, ps = rc.terrain_load_file(f)
        ty2, ps2 = da.terrain_load_file(f)
        obj = om.load_json(f)
        assert ty == ty2 == obj.get("Type", "")
        assert ps.shape == ps2.shape and np.array_equal(ps, ps2), f
        for k, o in enumerate(obj["Params"]):
            assert np.array_equal(ps[k], np.array(om.terrain_params_from_json(o), np.float64)), f


# ----------------------------------------------------------------------------------------------------------------------
# cRand
def _minstd_canonical(seed, n):
    """std::default_random_engine = minstd_rand0 (x <- 16807 x mod 2^31-1, seed 0 -> 1); uniform_real_distribution<double>(0, 1) =
    generate_canonical<double, 53>: two draws, (x1 - 1 + (x2 - 1) * R) / R^2 with R = 2^31 - 2 (libstdc++ bits/random.tcc)."""
    m = 2147483647
    x = seed % m
    if x == 0:
        x = 1
    out = []
    R = float(m - 1)
    for _ in range(n):
        x = (16807 * x) % m; a = x
        x = (16807 * x) % m; b = x
        v = ((a - 1) + (b - 1) * R) / (R * R)
        out.append(v if v < 1.0 else np.nextafter(1.0, 0.0))
    return np.array(out)


@needs_lib
def test_crand_streams_are_libstdcxx_minstd():
    for seed in (0, 1, 7, 123456789, 4294967295):
        got = rc.rand_stream(seed, "double", 0.0, 1.0, 64)
        assert np.array_equal(got, _minstd_canonical(seed, 64)), seed
        got = rc.rand_stream(seed, "double", -2.5, 4.0, 16)
        exp = -2.5 + _minstd_canonical(seed, 16) * 6.5
        assert np.array_equal(got, exp)
    # RandInt(min, max) = min + gen() % (max - min) with uniform_int_distribution<int>(0, INT_MAX): one engine draw, shifted down by one
    m = 2147483647
    x = 1
    exp = []
    for _ in range(32):
        x = (16807 * x) % m
        exp.append(0 + (x - 1) % 3)
    # libstdc++ maps the engine range [1, m-1] onto [0, INT_MAX] with a rejection / scaling step that is the identity shift here
    got = rc.rand_stream(1, "int", 0, 3, 32)
    assert set(np.unique(got)) <= {0.0, 1.0, 2.0}
    assert abs(np.mean(rc.rand_stream(1, "int", 0, 3, 3000)) - 1.0) < 0.1
    assert abs(np.mean(rc.rand_stream(2, "coin", 0.5, 0, 4000)) - 0.5) < 0.05
    nrm = rc.rand_stream(3, "norm", 1.0, 2.0, 4000)
    assert abs(nrm.mean() - 1.0) < 0.15 and abs(nrm.std() - 2.0) < 0.15


# ----------------------------------------------------------------------------------------------------------------------
# arg files
@needs_ref
def test_every_reference_arg_file_parses_like_the_reference(om, da):
    """util/ArgParser.cpp:42-108 (tokeniser) + :131-150 (ParseString): every key of every shipped arg file."""
    files = sorted(glob.glob(os.path.join(REFERENCE, "args", "*.txt")))
    assert len(files) == 19
    for f in files:
        ref = rc.RefArgs(f)
        toks = om.parse_arg_file(f)
        d = om.args_to_dict(toks)
        keys = sorted({t[1:-1] for t in toks if len(t) >= 3 and t[0] == "-" and t[-1] == "="})
        assert keys, f
        argv = ["-arg_file=", f]
        _, n_prod = da.args_parse_string(argv, "scenario")
        assert ref.count() == len(toks) == n_prod - 2, f
        for k in keys + ["no_such_key"]:
            r = ref.string(k)
            p, _ = da.args_parse_string(argv, k)
            assert r == d.get(k) == p, (f, k, r, d.get(k), p)
        # typed accessors on the keys the engine consumes
        for k in ("num_update_steps", "num_sim_substeps", "num_threads", "tuple_buffer_size", "trainer_replay_mem_size"):
            if k in d:
                assert ref.int(k) == int(d[k])
        for k in ("world_scale", "exp_rate", "exp_temp", "exp_base_rate", "char_init_pos_x", "terrain_blend"):
            if k in d:
                assert ref.double(k) == float(d[k])


@needs_ref
def test_command_line_overrides_the_arg_file_like_the_reference(da):
    f = os.path.join(REFERENCE, "args", "opt_args_train_mace.txt")
    ref = rc.RefArgs(f, argv=["-num_threads=", "17", "-exp_rate=", "0.5"])
    assert ref.int("num_threads") == 17 and ref.double("exp_rate") == 0.5
    p, _ = da.args_parse_string(["-num_threads=", "17", "-exp_rate=", "0.5", "-arg_file=", f], "num_threads")
    assert p == "17"


# ----------------------------------------------------------------------------------------------------------------------
# kinematic tree + rigid-body-dynamics model
def _states(m, D, n, seed):
    rng = np.random.RandomState(seed)
    q0 = np.array(m.pose0[:D]); qd0 = np.array(m.vel0[:D])
    out = [(q0.copy(), qd0.copy()), (q0.copy(), np.zeros(D))]
    for _ in range(n):
        out.append((q0 + rng.uniform(-0.7, 0.7, D), qd0 + rng.uniform(-4, 4, D)))
    return out


@needs_ref
@pytest.mark.parametrize("name,arg", CHARS)
def test_kin_tree_tables_match_the_reference_loader(om, name, arg):
    """cKinTree::Load / LoadBodyDefs / PostProcessJointMat (anim/KinTree.cpp:118-160, 409-457, 990-1023) vs oracle/model.py's reading."""
    m, info = om.build_model(arg, REFERENCE)
    r = rc.RefChar(os.path.join(REFERENCE, info["args"]["character_file"]))
    assert (r.L, r.D) == (m.L, m.D) and abs(r.total_mass - sum(m.body_mass[:m.L])) < 1e-12
    j, b = r.tables()
    off = 0
    for k in range(m.L):
        assert int(j[k, 0]) == m.joint_type[k] and int(j[k, 1]) == m.parent[k]
        assert int(j[k, 2]) == off and int(j[k, 3]) == (3 if m.joint_type[k] == 1 else 1)
        off += int(j[k, 3])
        assert np.array_equal(j[k, 4:7], np.array(m.attach[k][:])), (name, k)
        assert j[k, 7] == m.lim_lo[k] and j[k, 8] == m.lim_hi[k]
        assert b[k, 0] == m.body_mass[k] and np.array_equal(b[k, 1:4], np.array(m.body_attach[k][:])) and b[k, 4] == m.body_theta[k]
        assert np.array_equal(b[k, 5:8], np.array(m.body_size[k][:]))


@needs_ref
@pytest.mark.parametrize("name,arg", CHARS)
def test_hinge_limit_reference_angles(om, da, name, arg):
    """mRefTheta (sim/SimCharacter.cpp:838-865) from the reference's own BodyJointTrans / ParentChildTrans / InvRigidMat / RotMatToAxisAngle == the oracle
    loader's closed form; the hinge limits then act on theta + ref_theta (sim/World.cpp:543-553, 624-626). Only the children of the dog's / goat's
    rotated root body (spine0, tail0, hip) carry a non-zero one."""
    m, info = om.build_model(arg, REFERENCE)
    r = rc.RefChar(os.path.join(REFERENCE, info["args"]["character_file"]))
    got = r.ref_theta()
    assert np.abs(got - np.array(m.ref_theta[:m.L])).max() < 1e-12
    nz = [j for j in range(m.L) if abs(got[j]) > 1e-9]
    assert nz == ([1, 9, 17] if name != "raptor" else [])
    if name != "raptor":
        assert abs(got[17] + 0.61) < 1e-12


@needs_ref
@pytest.mark.parametrize("name,arg", CHARS)
def test_rbd_model_matches_the_reference(om, name, arg):
    """cRBDModel::Update -> BuildMassMat (CRBA), BuildBiasForce (RNEA with BuildCjPlanar as shipped), CalcGravityForce, CalcCoM."""
    m, info = om.build_model(arg, REFERENCE)
    e = om.OracleEnv(m)
    r = rc.RefChar(os.path.join(REFERENCE, info["args"]["character_file"]))
    D, L = e.D, e.L
    for q, qd in _states(m, D, 40, 3):
        H, Cq, Ct, g = e.rbd(q, qd)
        o = r.rbd(q, qd)
        sH, sC, sg = np.abs(o["H"]).max(), max(np.abs(o["C"]).max(), 1.0), max(np.abs(o["grav"]).max(), 1.0)
        assert np.abs(H - o["H"]).max() < 1e-12 * sH
        assert np.abs(Cq - o["C"]).max() < 1e-12 * sC           # the shipped quirk is part of the reference's C
        assert np.abs(g - o["grav"]).max() < 1e-12 * sg
        # textbook bias == reference bias only when the root does not move (the quirk is a pure velocity-product term)
        if not np.any(qd[:3]):
            assert np.abs(Ct - o["C"]).max() < 1e-9 * sC
        # body COM positions / velocities: oracle kinematics vs the reference's cKinTree + Jacobian route
        e.set_pose_vel(q, qd)
        c, v, psi = e.bodies()
        bp, bt, jp, jt = r.kin_bodies(q)
        assert np.abs(c - bp[:, :2]).max() < 1e-12
        assert np.abs(np.angle(np.exp(1j * (psi - bt)))).max() < 1e-12
        assert np.abs(o["joint_pos"] - jp).max() < 1e-12
        mass = np.array(m.body_mass[:L])
        com = (mass[:, None] * c).sum(0) / mass.sum(); comv = (mass[:, None] * v).sum(0) / mass.sum()
        assert np.abs(com - o["com"][:2]).max() < 1e-12 and np.abs(comv - o["com_vel"][:2]).max() < 1e-11
        # per-body COM velocity: cKinTree::CalcWorldVel at the body attach point (what cSimCharacter::SetVel assigns, sim/SimCharacter.cpp:227-315)
        for k in (0, L // 2, L - 1):
            ba = np.array(m.body_attach[k][:]); ba[2] = 0
            rv = r.world_vel(q, qd, k, ba)
            assert np.abs(rv[:2] - v[k]).max() < 1e-11, (name, k)


@needs_ref
def test_inverse_dynamics_identity_on_the_reference(om):
    """tau = H a + C on the reference's own code (SolveInvDyna with an acceleration): ties its CRBA and RNEA together, and the oracle to both."""
    m, info = om.build_model("args/dog_slopes_mixed_args.txt", REFERENCE)
    e = om.OracleEnv(m)
    r = rc.RefChar(os.path.join(REFERENCE, info["args"]["character_file"]))
    rng = np.random.RandomState(5)
    for q, qd in _states(m, e.D, 6, 9):
        a = rng.uniform(-20, 20, e.D)
        tau = r.inv_dyna(q, qd, a)
        H, Cq, _, _ = e.rbd(q, qd)
        assert np.abs(tau - (H @ a + Cq)).max() < 1e-10 * max(1.0, np.abs(tau).max())


# ----------------------------------------------------------------------------------------------------------------------
# frozen reference outputs (always run)
def test_oracle_and_product_match_the_frozen_reference_outputs(om, da):
    g = np.load(os.path.join(GOLDEN, "ref_golden.npz"))
    n_t = 0
    for key in g.files:
        if not key.startswith("terrain/"):
            continue
        _, ttype, pset, seed, width = key.split("/")
        prm = g["params/" + pset]
        ref = g[key]
        orc = om.terrain_build(om.TERRAIN_TYPES.index(ttype), prm, int(seed), float(width))
        prod, _ = da.terrain_build(ttype, prm, int(seed), float(width))
        assert np.array_equal(ref.view(np.uint32), orc.view(np.uint32)), key
        assert np.array_equal(ref.view(np.uint32), prod.view(np.uint32)), key
        n_t += 1
    assert n_t >= 28
    for name, arg in CHARS:
        m, _ = om.build_model(arg, REFDATA)
        e = om.OracleEnv(m)
        Q, QD = g["rbd/%s/q" % name], g["rbd/%s/qd" % name]
        for k in range(len(Q)):
            H, Cq, _, gr = e.rbd(Q[k], QD[k])
            for got, ref in ((H, g["rbd/%s/H" % name][k]), (Cq, g["rbd/%s/C" % name][k]), (gr, g["rbd/%s/grav" % name][k])):
                assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()), (name, k)
            e.set_pose_vel(Q[k], QD[k])
            c, v, _ = e.bodies()
            assert np.abs(c - g["rbd/%s/body_pos" % name][k][:, :2]).max() < 1e-12
