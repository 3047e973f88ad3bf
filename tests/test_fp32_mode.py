"""The opt-in fp32 mode (-physics_precision= f32): the SAME kernel source built with `real` = float (deepterrainrl_amd/lib/libdtrl_f32.so; Bullet's own
state is float, premake4.lua:115-124). Half the registers and half the LDS per env -> more wave slots per CU (profiles/r06_fp32_physics.txt). Never the headline: float
trajectories leave the fp64 oracle within a few env-steps (chaos), so this mode is held at DISTRIBUTION level against the fp64 build, and pointwise only over a short
horizon and against its own lane-loop check build."""
import os

import numpy as np
import pytest

import test_host_and_emul as T
from conftest import REFDATA, EMUL_LIB_F32, EmulScenario, dog_policy, emul_f32_scenario, trained_policy


def _stats(b, frames):
    b.RunFrames(frames)
    st = b.EvalStats(); q, qd = b.PoseVel()
    assert np.isfinite(q).all() and np.isfinite(qd).all()
    steps = b.num_envs * frames * 20.0
    return dict(falls_k=1000.0 * st["resets"] / steps, cycles=st["cycles"], avg_dist=st["avg_dist"], episodes=st["episodes"], steps=steps)


def check_distribution(a, b, tag, falls_abs=0.06, falls_rel=0.25):
    """fp32 run `a` against fp64 run `b` of the same envs: gait cycles 3 %, falls within max(25 %, 0.06 per 1000 env-steps), and -- where both runs logged at least 50
    episodes (a trained policy falls a handful of times per run) -- the mean episode distance 15 %"""
    assert abs(a["cycles"] - b["cycles"]) <= 0.03 * b["cycles"], (tag, a, b)
    assert abs(a["falls_k"] - b["falls_k"]) <= max(falls_abs, falls_rel * b["falls_k"]), (tag, a, b)
    if min(a["episodes"], b["episodes"]) >= 50:
        assert abs(a["avg_dist"] - b["avg_dist"]) <= 0.15 * b["avg_dist"], (tag, a, b)


def test_fp32_check_build_loads_and_libraries_refuse_the_other_precision(da):
    b = emul_f32_scenario("args/sim_dog_args.txt", 1, data_root=REFDATA, extra_args={"physics_precision": "f32"})
    assert "fp32" in b._lib.dtrl_version().decode()
    b.StepUpdates(1); b.close()
    with pytest.raises(da.DtrlError, match="physics_precision"):
        emul_f32_scenario("args/sim_dog_args.txt", 1, data_root=REFDATA, extra_args={"physics_precision": "f64"})
    with pytest.raises(da.DtrlError, match="physics_precision"):
        EmulScenario("args/sim_dog_args.txt", 1, data_root=REFDATA, extra_args={"physics_precision": "f32"})
    with pytest.raises(da.DtrlError, match="physics_precision"):
        EmulScenario("args/sim_dog_args.txt", 1, data_root=REFDATA, extra_args={"physics_precision": "half"})


def test_fp32_short_horizon_vs_oracle(da, om):
    """60 env-steps (300 substeps) of the flat-ground dog and raptor in float against the fp64 oracle: float rounding (6e-8) through the stiff contact rows -- |dq| < 1e-4,
    |dqd| < 2e-2 (observed 2e-5 / 4e-3); same ABI (doubles in, doubles out), policy state readable"""
    for arg in ("args/sim_dog_args.txt", "args/sim_raptor_args.txt"):
        m, _ = om.build_model(arg, REFDATA); e = om.OracleEnv(m, terrain_seed=2)
        b = emul_f32_scenario(arg, 1, data_root=REFDATA, extra_args={"terrain_seed": 2})
        for k in range(60):
            b.StepUpdates(1); e.step(1)
        q, qd = b.PoseVel(); qo, qdo = e.pose_vel()
        assert np.abs(q[0] - qo).max() < 1e-4 and np.abs(qd[0] - qdo).max() < 2e-2, (arg, np.abs(q[0] - qo).max(), np.abs(qd[0] - qdo).max())
        assert np.array_equal(b.Contacts()[0], np.array(e.contacts()))
        ps = b.RecordPoliState(); assert ps.shape[1] == b.S and np.isfinite(ps).all()


@pytest.mark.parametrize("arg,which", [("args/dog_slopes_mixed_args.txt", "dog_trained"), ("args/raptor_narrow_gaps_args.txt", "raptor_xavier"), ("args/goat_cliffs_args.txt", "goat_xavier")])
def test_fp32_distribution_level_parity_check_builds(da, om, arg, which):
    """fp32 vs fp64 lane-loop builds, 160 envs x 160 frames from the same seeds and policy: falls, gait cycles, episode distance (the full-size twin runs on the GPU)"""
    pol = {"dog_trained": lambda: trained_policy(om, "dog"), "raptor_xavier": lambda: T.raptor_policy(om), "goat_xavier": lambda: dog_policy(om)}[which]()
    out = []
    for make in (emul_f32_scenario, EmulScenario):
        b = make(arg, 160, data_root=REFDATA, extra_args={"terrain_seed": 900})
        b.SetPolicy(pol[1], *pol[2:])
        out.append(_stats(b, 160)); b.close()
    print(which, "fp32", out[0], "fp64", out[1])
    check_distribution(out[0], out[1], which)


# ---- GPU ----
@pytest.mark.gpu
def test_gpu_fp32_library_vs_its_check_build(da, om):
    """libdtrl_f32.so (HIP, gfx950) against the lane-loop build of the same fp32 source: 64 dogs on slopes_mixed with the policy forward in float (fp32 matrix pipe:
    v_mfma_f32_16x16x4_f32, result rows 4 g + r instead of 4 r + g), 4 outer frames = 400 substeps; agreement at float rounding amplified by the contact rows."""
    pol = dog_policy(om)
    kw = dict(data_root=REFDATA, extra_args={"terrain_seed": 77, "physics_precision": "f32"})
    b = da.BatchScenario("args/dog_slopes_mixed_args.txt", 64, **kw); c = emul_f32_scenario("args/dog_slopes_mixed_args.txt", 64, **kw)
    assert "fp32" in b._lib.dtrl_version().decode()
    maps = open("/proc/self/maps").read(); assert "libdtrl_f32.so" in maps
    for s in (b, c):
        s.SetPolicy(pol[1], *pol[2:])
    worst = 0.0
    for f in range(4):
        b.Update(); c.Update()
        qb, qdb = b.PoseVel(); qc, qdc = c.PoseVel()
        worst = max(worst, np.abs(qb - qc).max())
        assert np.array_equal(b.Contacts(), c.Contacts()) or f > 1
    print("HIP fp32 vs lane-loop fp32, 64 envs x 400 substeps: max |dq| %.2e, median %.2e" % (worst, np.median(np.abs(qb - qc).max(1))))
    assert np.median(np.abs(qb - qc).max(1)) < 1e-3 and np.isfinite(qb).all()
    ps_b, ps_c = b.RecordPoliState(), c.RecordPoliState()
    assert np.median(np.abs(ps_b - ps_c).max(1)) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("arg,which,n", [("args/dog_slopes_mixed_args.txt", "dog_xavier", 4096), ("args/dog_slopes_mixed_args.txt", "dog_trained", 4096),
                                         ("args/raptor_narrow_gaps_args.txt", "raptor_xavier", 8192), ("args/raptor_narrow_gaps_args.txt", "raptor_trained", 8192)])
def test_gpu_fp32_distribution_level_parity_full_width(da, om, arg, which, n):
    """-physics_precision= f32 against the fp64 product at the BASELINE widths, 150 frames from the same seeds: gait cycles 3 %, falls max(25 %, 0.06 / 1000 env-steps),
    episode distance 15 %."""
    pol = {"dog_xavier": lambda: dog_policy(om), "raptor_xavier": lambda: T.raptor_policy(om), "dog_trained": lambda: trained_policy(om, "dog"),
           "raptor_trained": lambda: trained_policy(om, "raptor")}[which]()
    out = []
    for prec in ("f32", "f64"):
        b = da.BatchScenario(arg, n, data_root=REFDATA, extra_args={"terrain_seed": 900, "physics_precision": prec})
        b.SetPolicy(pol[1], *pol[2:])
        out.append(_stats(b, 150)); b.close()
    print(which, "fp32", out[0], "fp64", out[1])
    check_distribution(out[0], out[1], which)
