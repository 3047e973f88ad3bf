"""-m gpu: the parity tests proper. Everything goes through the C ABI of libdtrl.so (HIP, gfx950) and is checked against the
CPU oracle on the same seeded inputs, against the committed golden vectors, and -- at BASELINE.json's full sizes -- through
size-independent properties (determinism, shard invariance, finiteness, reset accounting)."""
import os

import numpy as np
import pytest

import test_host_and_emul as T
from conftest import REFDATA, GOLDEN, HIP_LIB, dog_policy, pin_to_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def hip_batch(monkeypatch):
    def make(da, arg, n, **extra):
        return da.BatchScenario(arg, n, data_root=REFDATA, extra_args=extra)   # product path: libdtrl.so on cuda:0
    monkeypatch.setattr(T, "batch", make)
    import deepterrainrl_amd
    monkeypatch.setattr(T, "Scenario", deepterrainrl_amd.BatchScenario)


def test_native_library_is_loaded(da):
    b = da.BatchScenario("args/sim_dog_args.txt", 1, data_root=REFDATA)
    b.StepUpdates(1)
    maps = open("/proc/self/maps").read()
    assert "libdtrl.so" in maps and "libamdhip64" in maps
    assert "hip" in da.version() or "dtrl" in da.version()


def test_config0_flat_1200_substeps(da, om):
    T.test_kernel_math_vs_oracle_flat_1200_substeps(da, om)


def test_ground_indices_bit_exact(da, om):
    T.test_ground_bit_exact_vs_oracle(da, om, "args/dog_slopes_mixed_args.txt", 17)
    T.test_ground_bit_exact_vs_oracle(da, om, "args/dog_narrow_gaps_args.txt", 9)


def test_poli_eval_resets(da, om):
    T.test_poli_eval_with_policy_and_resets(da, om)


def test_nn_forward(da, om):
    T.test_nn_forward_golden(da, om)


def test_policy_output_all_net_families(da, om):
    T.test_policy_output_vs_oracle_forward(da, om)


def test_policy_forward_other_shapes(da, om, tmp_path):
    T.test_policy_forward_other_conv_shapes_and_nonzero_biases(da, om, tmp_path)


def test_exploration_tuples(da, om):
    T.test_exploration_tuples_vs_oracle_and_golden(da, om)


def test_goat(da, om):
    T.test_goat_cliffs_config(da, om)


def test_shard_invariance_small(da, om):
    T.test_shard_invariance(da, om)


def test_cacla_action_selection(da, om):
    T.test_cacla_action_selection_and_tuples_vs_oracle(da, om)


def test_raptor_cacla_action_selection(da, om, tmp_path):
    T.test_raptor_cacla_action_selection_and_tuples_vs_oracle(da, om, tmp_path)


def test_q_head_action_selection(da, om):
    T.test_q_head_action_selection_and_tuples_vs_oracle(da, om)


def test_perturbation_force(da, om):
    T.test_perturbation_force_vs_oracle(da, om)
    T.test_apply_rand_force_is_seeded_and_bounded(da, om)


def test_poli_eval_recorders(da, om, tmp_path):
    T.test_poli_eval_recorders_frame_polling_equals_env_step_polling(da, om, tmp_path)
    T.test_poli_eval_recorders_across_resets(da, om, tmp_path)


def test_set_pose_reset(da, om):
    T.test_set_pose_vel_and_reset_roundtrip(da, om)


def test_contact_cache_and_model_switches(da, om):
    T.test_contact_cache_is_part_of_the_state_and_model_switches(da, om)


def test_row_cap_prone_character(da, om):
    T.test_row_cap_prone_character_vs_oracle(da, om)


def test_edge_inputs_and_error_paths(da, om):
    T.test_edge_inputs_and_error_paths(da, om)


def test_scale_file_roundtrip(da, om, tmp_path):
    T.test_scale_file_roundtrip(da, om, tmp_path)


@pytest.mark.parametrize("terrain", ["gaps", "mixed", "narrow_gaps", "cliffs_rugged"])
def test_shipped_terrain_types(da, om, terrain):
    T.test_every_shipped_terrain_type_vs_oracle(da, om, terrain)


def test_terrain_param_lerp_curriculum(da, om, tmp_path, monkeypatch):
    monkeypatch.setattr(T, "Scenario", da.BatchScenario)
    T.test_terrain_param_lerp_curriculum(da, om, tmp_path)


def test_raptor_flat_and_narrow_gaps(da, om):
    """BASELINE config 2 (raptor, different KinTree topology, D = 21 kernel instantiation)."""
    T.test_raptor_flat_1200_substeps_vs_oracle(da, om)
    T.test_raptor_narrow_gaps_with_policy(da, om)


def test_raptor_full_size_8192(da, om):
    """BASELINE config 2 at full size: 8192 raptor envs on narrow_gaps, determinism + finiteness + episode accounting."""
    pol = T.raptor_policy(om)
    def run():
        b = T.batch(da, "args/raptor_narrow_gaps_args.txt", 8192, terrain_seed=5)
        b.SetPolicy(pol[1], *pol[2:])
        b.RunFrames(20)
        return b
    a = run(); c = run()
    qa, qda = a.PoseVel(); qc, qdc = c.PoseVel()
    assert np.isfinite(qa).all() and np.array_equal(qa, qc) and np.array_equal(qda, qdc)
    st = a.EvalStats()
    assert st["cycles"] >= 8192 * 2 and st["episodes"] == st["resets"]


def test_config1_slopes_mixed_1200_substeps_64_envs(da, om):
    """BASELINE config 1 at reduced width for the oracle side: 64 envs, dog + slopes_mixed + MACE forward, 12 frames = 1200
    substeps, per-frame |dq| < 1e-4 (north-star bound; observed ~1e-10)."""
    m, info = om.build_model("args/dog_slopes_mixed_args.txt", REFDATA)
    pol = dog_policy(om)
    n = 64
    b = T.batch(da, "args/dog_slopes_mixed_args.txt", n, terrain_seed=1000)
    b.SetPolicy(pol[1], *pol[2:])
    es = [om.OracleEnv(m, terrain_seed=1000 + i, rng_seed=0, env_id=i, policy=pol) for i in range(n)]
    worst = 0
    for f in range(12):
        b.Update()
        for e in es:
            e.update()
        q, qd = b.PoseVel()
        for i, e in enumerate(es):
            qo, qdo = e.pose_vel()
            worst = max(worst, np.abs(q[i] - qo).max())
            assert np.abs(q[i] - qo).max() < 1e-6 and np.abs(qd[i] - qdo).max() < 1e-6   # north-star bound 1e-4; observed ~1e-10
    print("config1 64 envs x 1200 substeps: max |dq| = %.3e" % worst)


FULL_WIDTH = [  # tag, arg file, envs (the width BASELINE.json states for the config), policy, terrain seed 0
    ("config1_dog_slopes_mixed_4096_xavier", "args/dog_slopes_mixed_args.txt", 4096, "dog_xavier", 1000),
    ("config2_raptor_narrow_gaps_8192_xavier", "args/raptor_narrow_gaps_args.txt", 8192, "raptor_xavier", 5000),
    ("config1_dog_slopes_mixed_4096_trained", "args/dog_slopes_mixed_args.txt", 4096, "dog_trained", 1000),
    ("config2_raptor_narrow_gaps_8192_trained", "args/raptor_narrow_gaps_args.txt", 8192, "raptor_trained", 5000),
    # configs[4]'s scene at its per-GPU width (65 536 goats over 8 GPUs): goat + cliffs_rugged, one substep of 1/600 s per env-step
    ("config4_goat_cliffs_8192_xavier", "args/goat_cliffs_args.txt", 8192, "goat_xavier", 9000),
    ("config4_goat_cliffs_8192_trained", "args/goat_cliffs_args.txt", 8192, "goat_trained", 9000),
]


@pytest.mark.parametrize("run", FULL_WIDTH, ids=[r[0] for r in FULL_WIDTH])
def test_config_full_width_1200_substeps(da, om, run):
    """VERDICT r5 #1a: pointwise HIP-vs-oracle parity at the widths BASELINE configs[1] / configs[2] state (4096 dogs, 8192 raptors), under the seeded xavier weights AND
    under the policies trained through the engine (long contact-rich episodes without falls). Every env against its own free-running oracle env (all host cores):
    phase A = 12 frames = 1200 substeps FREE-RUNNING, max |dq|, |dqd| < 1e-4 for EVERY env (north star); phase B = 36 more frames followed frame by frame from the
    oracle's state (through stumbles, prone characters at the row caps, falls, resets, terrain slides). The record (distribution, how many envs went through R >= 16,
    a row cap, link--link rows, a reset) goes to gpurun_out/full_width_parity/<tag>.txt -> profiles/r06_full_width_parity.txt."""
    from conftest import trained_policy
    tag, arg, n, which, seed = run
    pol = {"dog_xavier": lambda: dog_policy(om), "raptor_xavier": lambda: T.raptor_policy(om),
           "dog_trained": lambda: trained_policy(om, "dog"), "raptor_trained": lambda: trained_policy(om, "raptor"),
           "goat_xavier": lambda: dog_policy(om), "goat_trained": lambda: trained_policy(om, "goat")}[which]()
    r = T.run_full_width_parity(da, om, arg, n, pol, seed, free_frames=12, forced_frames=36, label=tag)
    from conftest import REPO
    out = os.path.join(REPO, "gpurun_out", "full_width_parity")
    try:
        os.makedirs(out, exist_ok=True)
        open(os.path.join(out, tag + ".txt"), "w").write(r["text"] + "\n")
    except OSError:
        pass
    T.check_full_width(r, min_tracked=0.8 if "goat" in tag else 0.85, min_within6=0.9)   # (observed: dog 0.92 / 1.00, raptor 0.96 / 0.99, goat 0.98 / 0.86 -- one substep of 1/600 s per env-step is the stiffest of the three)


def test_full_size_4096_properties(da, om):
    """BASELINE config 1 at full size (4096 envs): determinism across two batches, shard invariance (2 x 2048 with global
    offsets == 1 x 4096), finite state, resets accounted, terrain indices valid."""
    pol = dog_policy(om)
    def run(n, off, frames):
        b = T.batch(da, "args/dog_slopes_mixed_args.txt", n, terrain_seed=7, global_env_offset=off)
        b.SetPolicy(pol[1], *pol[2:])
        b.RunFrames(frames)
        return b
    frames = 30
    full = run(4096, 0, frames); again = run(4096, 0, frames)
    qf, qdf = full.PoseVel(); qa, qda = again.PoseVel()
    assert np.isfinite(qf).all() and np.isfinite(qdf).all()
    assert np.array_equal(qf, qa) and np.array_equal(qdf, qda)                       # bitwise deterministic
    lo = run(2048, 0, frames); hi = run(2048, 2048, frames)
    assert np.array_equal(qf[:2048], lo.PoseVel()[0]) and np.array_equal(qf[2048:], hi.PoseVel()[0])
    st = full.EvalStats()
    assert st["cycles"] >= 4096 * 2 and st["episodes"] == st["resets"]
    assert (np.abs(qf[:, 1]) < 50).all() and (qf[:, 0] > -25).all()
    h, seg, i, j = full.SampleGround(4095, np.linspace(qf[4095, 0] - 1, qf[4095, 0] + 10, 50))
    assert np.isfinite(h).all() and ((j - i) >= 0).all() and ((j - i) <= 1).all()
    ms, n = full.KernelTimeMs()
    assert n >= frames and ms > 0


@pytest.mark.parametrize("arg,n,frames", [("args/dog_slopes_mixed_args.txt", 256, 30), ("args/raptor_narrow_gaps_args.txt", 192, 40),
                                          ("args/opt_args_train_mace.txt", 128, 40), ("args/goat_cliffs_args.txt", 128, 40)])
def test_fast_kernel_equals_reference_kernel_bitwise(da, om, monkeypatch, arg, n, frames):
    """The register-resident gfx950 kernels (fast<23> dog/goat, fast<21> raptor) and the LDS-phase reference kernel (DTRL_KERNEL=ref)
    perform the same arithmetic in the same order: falls, resets, policy forwards, exploration and tuples included, everything must
    agree bit for bit."""
    raptor = "raptor" in arg
    pol = T.raptor_policy(om) if raptor else dog_policy(om)
    def run(kernel):
        if kernel:
            monkeypatch.setenv("DTRL_KERNEL", kernel)
        else:
            monkeypatch.delenv("DTRL_KERNEL", raising=False)
        b = T.batch(da, arg, n, terrain_seed=77)
        b.SetPolicy(pol[1], *pol[2:])
        b.RunFrames(frames)
        rows, flags, ids = b.DrainTuples()
        o = np.lexsort((np.arange(len(ids)), ids))
        return b.PoseVel(), b.Torques(), b.EvalStats(), b.Ctrl(), rows[o], flags[o]
    (qf, qdf), (tcf, taf), sf, cf, rf, ff = run(None)
    (qr, qdr), (tcr, tar), sr, cr, rr, fr = run("ref")
    assert np.array_equal(qf, qr) and np.array_equal(qdf, qdr) and np.array_equal(tcf, tcr) and np.array_equal(taf, tar)
    assert sf == sr and all(np.array_equal(a, b) for a, b in zip(cf, cr)) and sf["cycles"] > n
    assert np.array_equal(rf, rr) and np.array_equal(ff, fr)
    if "train" in arg:
        assert len(rf) > n // 2


def test_env_groups_pipelined_run_frames(da, om, monkeypatch):
    T.test_env_groups_pipelined_run_frames_equals_stepwise(da, om, monkeypatch)


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the driver's contract keys, the roofline object and (unless skipped) the CPU baseline."""
    import json, subprocess, sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "2", "--envs-per-gpu", "512", "--cpu-frames", "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and d["value"] > 1e5
    # steady-state protocol: a pre-roll of >= 60 frames independent of --warmup, R >= 3 windows of exactly --steps frames, value = the median window
    assert d["preroll"] >= 60 and d["repeats"] >= 3 and d["value_min"] <= d["value"] <= d["value_max"]
    assert abs(d["ms_per_step"] - d["window_s"]["median"] / 3 * 1e3) < 1e-9 and d["timed_window"]["frames"] == 3 * d["repeats"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel_launches"] == 3 * d["repeats"] and rf["kernel_avg_ms"] > 0
    ex = d["exchange"]
    assert "error" not in ex and ex["tuple_block_rows"] == 64 and ex["dropped_tuples"] == 0 and ex["env_steps_per_s"] > 1e5
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb


def test_bench_trained_policy_leg_and_policy_files():
    """The policies tools/learn_curve.py trained THROUGH this engine (tests/golden/policies: Caffe HDF5 + _scale.txt, written by the package's own writer) load through
    cNeuralNet::LoadModel's route and carry the dog across slopes_mixed: the bench's side figure under the trained policy has a fall rate far below the headline's
    (seeded xavier weights), and the headline itself is untouched by the leg."""
    import json, subprocess, sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--exchange-steps", "0", "--no-rccl-leg"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    tp = d["trained_policy"]
    assert "error" not in tp and tp["env_steps_per_s"] > 1e6 and tp["policy"].endswith("dog_mace3_slopes_mixed_model.h5")
    assert tp["resets_per_frame"] < 0.2 * d["timed_window"]["resets_per_frame"], (tp["resets_per_frame"], d["timed_window"]["resets_per_frame"])
    assert d["window_s"]["total"] >= 4.5 and len(d["window_s"]["each"]) == d["repeats"]     # the timed region is long enough for an SMI sample to land in it


def test_bench_under_a_launcher_with_the_rccl_path_on_one_rank():
    """The N > 1 code path as far as one GPU can take it: bench.py started by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
    launcher's environment, as the driver starts it) with a one-rank RCCL group (DTRL_FORCE_COLLECTIVES=1): process-group start-up, the barrier / all-reduce of
    the timed windows, the gather of the packed tuple block and the policy broadcast all run through RCCL; one JSON line from rank 0."""
    import json, subprocess, sys
    from conftest import REPO
    env = dict(os.environ, DTRL_FORCE_COLLECTIVES="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--envs-per-gpu", "512", "--exchange-steps", "40", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl"] and d["rccl"]["ranks"] == 1 and d["value"] > 1e5
    ex = d["exchange"]
    assert "error" not in ex and "RCCL" in ex["collective"] and ex["tuples"] > 0 and ex["dropped_tuples"] == 0
    assert ex["collective_bytes_per_frame"]["sent_per_rank"] == ex["tuple_block_bytes"] > 0


def test_fsm_controllers_locomote_at_scale(da, om):
    """Behavioural sanity at batch scale (no network, shipped FSM parameters): the bounding dog and the running raptor make steady
    forward progress on flat ground in every env -- the same check tests/test_oracle_kat.py applies to the oracle's single env."""
    for arg, min_speed in (("args/sim_dog_args.txt", 3.0), ("args/sim_raptor_args.txt", 2.5)):
        b = T.batch(da, arg, 512, terrain_seed=1)
        b.RunFrames(300)                                   # 10 s of simulated time
        q, qd = b.PoseVel()
        assert np.all(np.isfinite(q)) and np.all(np.isfinite(qd))
        speed = q[:, 0] / 10.0
        assert speed.min() > min_speed, (arg, speed.min(), speed.mean())
        assert np.abs(speed - speed.mean()).max() < 1e-9   # flat ground, identical envs: identical trajectories in all 512 lanes of work


def test_raptor_gravity_comp_and_virtual_forces_on_the_gpu(da, om, tmp_path):
    """The TopoRaptor fast kernel with EnableGravityCompensation / EnableVirtualForces switched ON (the shipped raptor.txt has both off, so
    the weighted contact-basis least squares and the stance / swing hip coupling would otherwise never run on the device)."""
    T.test_raptor_gravity_comp_and_virtual_forces_paths(da, om, tmp_path)


@pytest.mark.parametrize("arg,n,frames", [("args/dog_slopes_mixed_args.txt", 8, 60), ("args/raptor_narrow_gaps_args.txt", 8, 60)])
def test_record_poli_state_fp64_every_cycle(da, om, arg, n, frames):
    """cNNController::RecordPoliState (ParseGround + BuildPoliState, raptor: stance-mirrored) in full fp64 at EVERY gait cycle of every env, not
    through the float32 tuple rows: whenever product and oracle start a cycle on the same env-step, the 283 / 275 features must agree."""
    raptor = "raptor" in arg
    m, _ = om.build_model(arg, REFDATA)
    pol = T.raptor_policy(om) if raptor else dog_policy(om)
    b = T.batch(da, arg, n, terrain_seed=500)
    b.SetPolicy(pol[1], *pol[2:])
    es = [om.OracleEnv(m, terrain_seed=500 + i, rng_seed=0, env_id=i, policy=pol) for i in range(n)]
    steps = frames * 20
    compared = mirrored = 0
    prev_cyc = np.zeros(n, np.int64); prev_cyc_o = np.zeros(n, np.int64)
    synced = np.ones(n, bool)
    for k in range(steps):
        b.StepUpdates(1)
        for e in es:
            e.step(1)
        cyc = b.CycleInfo()[0]
        q, qd = b.PoseVel()
        ps = None
        for i, e in enumerate(es):
            co = e.stats()["cycles"]
            new_p, new_o = cyc[i] != prev_cyc[i], co != prev_cyc_o[i]
            prev_cyc[i] = cyc[i]; prev_cyc_o[i] = co
            qo, qdo = e.pose_vel()
            dq, dqd = np.abs(q[i] - qo).max(), np.abs(qd[i] - qdo).max()
            if dq > 1e-7:
                synced[i] = False          # StepUpdates never resets: after the first fall / tumble this env is out of the comparison
            if not synced[i]:
                continue
            assert new_p == new_o, (k, i)
            if new_p:
                if ps is None:
                    ps = b.RecordPoliState()
                so = e.poli_state()
                # the features are a smooth function of (q, qd): fp64-tight where the states are, never looser than 50x the state difference
                assert np.abs(ps[i] - so).max() < 1e-10 * max(1.0, np.abs(so).max()) + 50 * (dq + dqd), (k, i, np.abs(ps[i] - so).max(), dq, dqd)
                compared += 1
        if k % 20 == 19:
            pin_to_oracle(b, es, tol=1e-7)   # compared first, then every env that is still with its oracle env starts the next 20 env-steps from a common state (conftest.pin_to_oracle)
    assert compared >= 3 * n, compared


def test_distribution_level_parity_4096_vs_oracle(da, om):
    """Beyond the chaos horizon trajectories cannot be compared one by one, distributions can: 4096 HIP envs vs 1024 oracle envs (same
    workload: BASELINE configs[1], different terrain seeds), 300 outer frames. Episode length, distance per episode, cycles per env-step
    and reset rate must agree within sampling error (4 sigma of the smaller sample + 3 %)."""
    arg = "args/dog_slopes_mixed_args.txt"
    m, _ = om.build_model(arg, REFDATA)
    pol = dog_policy(om)
    frames = 300
    b = T.batch(da, arg, 4096, terrain_seed=9000)
    b.SetPolicy(pol[1], *pol[2:])
    b.RunFrames(frames)
    st = b.EvalStats()
    d, ids = b.GetDistLog()
    n_o = 1024
    o = om.batch_eval(m, n_o, os.cpu_count() or 8, frames, terrain_seed0=700000, rng_seed=0, env_id0=0, policy=pol)
    steps_p, steps_o = 4096.0 * frames * 20, o["env_steps"]
    # resets per env-step (Poisson-ish counts)
    rp, ro = st["resets"] / steps_p, o["resets"] / steps_o
    sig = np.sqrt(max(o["resets"], 1.0)) / steps_o
    assert o["resets"] > 30 and abs(rp - ro) < 4 * sig + 0.03 * ro, (rp, ro, sig)
    cp, co = st["cycles"] / steps_p, o["cycles"] / steps_o
    assert abs(cp - co) < 0.02 * co, (cp, co)
    # distance per episode
    mp, mo = d.mean(), o["dist_sum"] / o["episodes"]
    so = np.sqrt(max(o["dist_sq_sum"] / o["episodes"] - mo * mo, 0.0) / o["episodes"])
    assert len(d) == st["episodes"] and abs(mp - mo) < 4 * so + 0.03 * abs(mo), (mp, mo, so)
    # episode length = env-steps per reset
    lp, lo = steps_p / max(st["resets"], 1), steps_o / max(o["resets"], 1)
    assert abs(lp - lo) < (4 / np.sqrt(max(o["resets"], 1.0)) + 0.03) * lo, (lp, lo)
    print("distribution parity: resets/env-step %.3e vs %.3e, cycles/env-step %.4e vs %.4e, dist/episode %.3f vs %.3f (+-%.3f)" % (rp, ro, cp, co, mp, mo, so))


# ---- boundary behaviour through libdtrl.so (twins of tests/test_boundary.py) ----
import test_boundary as TB


@pytest.fixture
def hip_boundary(monkeypatch, da):
    monkeypatch.setattr(TB, "Scenario", da.BatchScenario)


def test_dist_log_on_the_gpu(da, om, tmp_path, hip_boundary):
    TB.test_dist_log_avg_dist_and_output_results(da, om, tmp_path)


def test_tuple_ring_overflow_counted_on_the_gpu(da, om, hip_boundary):
    TB.test_tuple_ring_overflow_is_counted_never_silent(da, om)


def test_env_id_validation_on_the_gpu(da, hip_boundary):
    TB.test_env_id_lists_are_validated(da)


def test_policy_hand_over_during_a_frame_on_the_gpu(da, om):
    """the deferred dtrl_set_policy_device (second weight buffer, switched in by the next launch) while a frame is REALLY in flight: same rollout as the waiting form"""
    import torch
    import test_boundary as TB
    dev = torch.device("cuda", 0)
    def to_dev(w):
        t = torch.from_numpy(w).to(dev); torch.cuda.synchronize()
        return (t, t.data_ptr())
    side = torch.cuda.Stream(device=dev)
    TB.run_policy_hand_over_during_a_frame(da.BatchScenario, om, to_dev=to_dev, n_envs=256, frames=40, stream_ptr=side.cuda_stream)


@pytest.mark.gpu
def test_device_resident_tuple_drain_and_policy_hand_over(da, om):
    """dtrl_drain_tuples_device / dtrl_set_policy_device with torch CUDA tensors: identical rows and rollout to the host-pointer calls."""
    import torch
    pol = dog_policy(om)
    args = dict(terrain_seed=31, rand_seed=2)
    a = da.BatchScenario("args/opt_args_train_mace.txt", 64, data_root=REFDATA, extra_args=args)
    b = da.BatchScenario("args/opt_args_train_mace.txt", 64, data_root=REFDATA, extra_args=args)
    a.SetPolicy(pol[1], *pol[2:])
    dev = torch.device("cuda", 0)
    w = torch.from_numpy(np.ascontiguousarray(pol[1], np.float32)).to(dev)
    nv = [torch.from_numpy(np.ascontiguousarray(x, np.float64)).to(dev) for x in pol[2:]]
    torch.cuda.synchronize()
    b.SetPolicyDevice(w.data_ptr(), w.numel(), *[x.data_ptr() for x in nv])
    cap = 256
    rows = torch.zeros((cap, b.W), dtype=torch.float32, device=dev); fl = torch.zeros(cap, dtype=torch.int32, device=dev); ids = torch.zeros(cap, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    n_tot = 0
    for f in range(60):
        a.Update(); b.Update()
        ra, fa, ia = a.DrainTuples()
        nb = b.DrainTuplesDevice(rows.data_ptr(), fl.data_ptr(), ids.data_ptr(), cap)
        assert nb == len(ra)
        # ring order = completion order of the wavefronts (not deterministic across two batches): compare per env (an env emits at most one tuple per frame)
        rb, fb, ib = rows[:nb].cpu().numpy(), fl[:nb].cpu().numpy().astype(np.uint32), ids[:nb].cpu().numpy()
        oa, ob = np.argsort(ia, kind="stable"), np.argsort(ib, kind="stable")
        assert np.array_equal(rb[ob], ra[oa]) and np.array_equal(fb[ob], fa[oa]) and np.array_equal(ib[ob], ia[oa])
        n_tot += nb
    assert n_tot >= 64
    assert np.array_equal(a.PoseVel()[0], b.PoseVel()[0])


@pytest.mark.skipif(not os.path.exists(os.path.join(TB.SHIM_DIR, "drive_shim_hip")), reason="tests/shim/drive_shim_hip not built")
def test_shim_drives_the_hip_library(da, om, tmp_path):
    """include/BatchScenarioExp.h (compiled inside the reference's header tree, tests/shim/Makefile) linked against libdtrl.so: 100 frames on the GPU,
    same tuple buffers as the Python mirror."""
    lines, pol = TB._run_shim(os.path.join(TB.SHIM_DIR, "drive_shim_hip"), tmp_path, om)
    py_lines, py_total = TB._python_side(100, 12, pol, scenario=da.BatchScenario)
    TB._check_shim_output(lines, py_lines, py_total)


@pytest.mark.skipif(not os.path.exists(os.path.join(TB.SHIM_DIR, "drive_shim_eval_hip")), reason="tests/shim/drive_shim_eval_hip not built")
def test_poli_eval_shim_drives_the_hip_library(da, om, tmp_path):
    """include/BatchScenarioPoliEval.h (compiled inside the reference's header tree on the build box, linked against libdtrl.so) runs cOptScenarioPoliEval's
    pool protocol on the GPU -- SetRandSeed + Reset, the EvalHelper loop over GetNumEpisodes / GetNumCycles / GetAvgDist / ResetAvgDist, GetDistLog,
    OutputResults -- and reports the same folds, dist log and results line as the Python mirror over the same library."""
    TB.run_eval_shim(os.path.join(TB.SHIM_DIR, "drive_shim_eval_hip"), tmp_path, om, da.BatchScenario, pool=64, max_episodes=200)


def _obb_overlap(c0, a0, h0, c1, a1, h1):
    """separating-axis test of two oriented boxes (centres c, angles a, half extents h), vectorised over a leading axis"""
    ok = np.ones(c0.shape[0], bool)
    d = c1 - c0
    for ang in (a0, a1):
        for k in range(2):
            ax = np.stack([np.cos(ang + k * np.pi / 2), np.sin(ang + k * np.pi / 2)], 1)
            r = 0
            for (aa, hh) in ((a0, h0), (a1, h1)):
                ux = np.stack([np.cos(aa), np.sin(aa)], 1); uy = np.stack([-np.sin(aa), np.cos(aa)], 1)
                r = r + hh[0] * np.abs((ux * ax).sum(1)) + hh[1] * np.abs((uy * ax).sum(1))
            ok &= np.abs((d * ax).sum(1)) <= r
    return ok


def test_same_group_nonadjacent_links_overlap_statistics(da, om):
    """in the reference links of one collision group that are not joined by a hinge DO collide with each other
    (sim/SimDog.cpp:73-81; only constraint-linked pairs are excluded, sim/World.cpp:626); Integrator v1 has no link-link contacts. This measures
    how often that matters over the bench workload: the fraction of (env, frame) samples of RUNNING (not fallen, not about to be reset)
    characters in which the boxes of such a pair intersect, per pair. Logged for DESIGN 4; the run fails if the gait itself needs them."""
    arg = "args/dog_slopes_mixed_args.txt"
    m, _ = om.build_model(arg, REFDATA)
    pol = dog_policy(om)
    n, frames = 512, 90
    b = T.batch(da, arg, n, terrain_seed=4242)
    b.SetPolicy(pol[1], *pol[2:])
    L = b.L
    half = np.array([[0.5 * m.body_size[j][0], 0.5 * m.body_size[j][1]] for j in range(L)])
    pairs = [(i, j) for i in range(L) for j in range(i + 1, L)
             if m.col_group[i] == m.col_group[j] != 0 and m.parent[j] != i and m.parent[i] != j]
    hits = np.zeros(len(pairs)); samples = 0
    prev_resets = b.CycleInfo()[1].copy()
    for f in range(frames):
        b.Update()
        c, v, a = b.LinkStates()
        resets = b.CycleInfo()[1]
        running = (resets == prev_resets) & ((b.Flags() & 1) == 0)     # not reset this frame, not fallen
        prev_resets = resets.copy()
        idx = np.nonzero(running)[0]
        samples += len(idx)
        for k, (i, j) in enumerate(pairs):
            hits[k] += _obb_overlap(c[idx, i], a[idx, i], half[i], c[idx, j], a[idx, j], half[j]).sum()
    rate = hits / max(samples, 1)
    worst = sorted(zip(rate, pairs), reverse=True)[:6]
    print("same-group non-adjacent box overlaps over %d running (env, frame) samples: any-pair upper bound %.4f; worst pairs %s" % (
        samples, rate.sum(), ", ".join("%d-%d: %.4f" % (p[0], p[1], r) for r, p in worst)))
    assert samples > 0.8 * n * frames
    assert rate.max() < 0.5


# ---- on-device terrain generation (tests/test_device_terrain.py holds the CPU twins and the description) ----
def test_device_terrain_rollout_windows_resets_and_dist_log(om):
    import test_device_terrain as D
    import deepterrainrl_amd as da_mod
    b = D.run_device_terrain_rollout(da_mod.BatchScenario, om, n=256, frames=150)
    assert b.EvalStats()["resets"] >= 20


def test_device_terrain_determinism_and_shard_invariance(om):
    import test_device_terrain as D
    import deepterrainrl_amd as da_mod
    D.run_determinism_and_shard_invariance(da_mod.BatchScenario, om)


def test_device_terrain_equals_the_lane_loop_build(om):
    """The GPU's windows against the same code run by the host compiler (tests/emul): identical records for the same global env ids."""
    import test_device_terrain as D
    import deepterrainrl_amd as da_mod
    from conftest import EmulScenario
    g = D.make(da_mod.BatchScenario, 16, terrain_seed=21, rand_seed=4); D.set_policy(g, om)
    c = D.make(EmulScenario, 16, terrain_seed=21, rand_seed=4); D.set_policy(c, om)
    for f in range(60):
        g.Update(); c.Update()
    qg, _ = g.PoseVel(); qc, _ = c.PoseVel()
    n_same = 0
    for e in range(16):
        if abs(qg[e, 0] - qc[e, 0]) < 1e-6:                                  # trajectories that have not yet diverged in the last bits (fast vs reference kernel)
            (wg, nbg), (wc, nbc) = g.GroundWindow(e), c.GroundWindow(e)
            assert nbg == nbc and all(np.array_equal(wg[s][2], wc[s][2]) and wg[s][0] == wc[s][0] for s in range(2))
            n_same += 1
    assert n_same >= 8


def test_device_generator_statistics_match_the_host_generator(om):
    import test_device_terrain as D
    import deepterrainrl_amd as da_mod
    D.run_generator_statistics(da_mod.BatchScenario, om, terrains=("slopes_mixed", "narrow_gaps"))


def test_device_terrain_user_reset_and_curriculum(om):
    import test_device_terrain as D
    import deepterrainrl_amd as da_mod
    D.run_user_reset_and_curriculum(da_mod.BatchScenario, om)


def test_nn_activation_recorder(da, om, tmp_path):
    T.test_nn_activation_recorder_vs_numpy_net_and_oracle_forward(da, om, tmp_path)


def test_packed_drain_equals_plain_drain(om):
    import test_boundary as B
    import deepterrainrl_amd as da_mod
    B.run_packed_drain_equals_plain_drain(da_mod.BatchScenario, om, to_ptr=lambda t: t.data_ptr(), read=lambda t: t.cpu().numpy())


def test_pipelined_drain_equals_sequential(om):
    import test_boundary as B
    import deepterrainrl_amd as da_mod
    B.run_pipelined_drain_equals_sequential(da_mod.BatchScenario, om, to_ptr=lambda t: t.data_ptr(), read=lambda t: t.cpu().numpy())


def test_step_poll_keeps_the_tuple_stream(om):
    """dtrl_step_poll on the MI355X: sometimes no group, sometimes one, sometimes both have finished when the poll comes (a short sleep stands for the trainer's work)"""
    import time
    import test_boundary as B
    import deepterrainrl_amd as da_mod
    k = [0]
    def work():
        k[0] += 1
        time.sleep(0.0004 * (k[0] % 4))
    B.run_step_poll_keeps_the_tuple_stream(da_mod.BatchScenario, om, n_envs=1536, frames=60, work=work)


def test_host_memory_tuple_ring_equals_the_device_ring(om):
    """-tuple_ring= host on the MI355X: the kernels write tuple rows into page-locked host memory (system-scope cursor atomic), the packed drain's kernels read
    them from there, the plain drain is a host memcpy -- same tuple stream as the device ring, 768 envs so that frames outlast the host"""
    import test_boundary as B
    import deepterrainrl_amd as da_mod
    B.run_pipelined_drain_equals_sequential(da_mod.BatchScenario, om, to_ptr=lambda t: t.data_ptr(), read=lambda t: t.cpu().numpy(), n_envs=768, cap=1536, frames=60,
                                            extra_b={"tuple_ring": "host"})


def test_pipelined_drain_equals_sequential_device_terrain(om):
    """-terrain_gen= device + tuple pipelining: the frame boundary is queued device work, the host never synchronises with a frame, so the
    drain of frame f's ring must wait ON THE DEVICE for frame f's launches (per-group frame marks). 768 envs in two groups: a frame lasts milliseconds
    while the host is back within microseconds -- without the marks the drain read torn rows and zeroed the cursor under the running kernel."""
    import test_boundary as B
    import deepterrainrl_amd as da_mod
    B.run_pipelined_drain_equals_sequential(da_mod.BatchScenario, om, to_ptr=lambda t: t.data_ptr(), read=lambda t: t.cpu().numpy(),
                                            n_envs=768, cap=1536, frames=60, extra={"terrain_gen": "device"})


def test_link_link_contacts(da, om):
    T.test_link_link_contacts_vs_oracle(da, om)


def test_product_vs_frozen_reference_lockstep_traces(da):
    T.test_product_vs_frozen_reference_lockstep_traces(da)


@pytest.mark.parametrize("run", T.CONFIG_RUNS, ids=[r[0] for r in T.CONFIG_RUNS])
def test_product_vs_frozen_reference_config_traces(da, om, run):
    """The HIP kernel on the MI355X against the compiled REFERENCE's frozen traces of the BASELINE scenes (tests/golden/ref_golden_configs.npz, made where
    /root/reference exists): dog + slopes_mixed + MACE net through two falls (configs[1], two seeds), raptor + narrow_gaps with the stance-mirrored
    state (configs[2]), goat + cliffs_rugged (configs[4]'s scene) -- torques, contact flags, FSM, actions, PD targets, policy states, counters, dist log."""
    info = T.run_product_vs_frozen_reference_config(da, om, *run, scenario=da.BatchScenario)
    print(run[0], info)
    T.report_tracked("hip:" + run[0], info)
    assert info["tracked"] >= (0.95 if run[0].endswith("_trained") else 0.6) * info["frames"] and info["cycles"] >= 5 and info["contact_frames"] >= 20, info
    assert info["resets_tracked"] >= min(1, T.CONFIG_MIN_RESETS[run[0]]), info


@pytest.mark.parametrize("run", T.TUPLE_RUNS, ids=[r[0] for r in T.TUPLE_RUNS])
def test_product_vs_frozen_reference_tuples(da, om, run):
    T.test_product_vs_frozen_reference_tuples(da, om, run, scenario=da.BatchScenario)


def test_sharded_rollout_pipelined_protocol_on_gpu(om):
    """ShardedRollout on the GPU (comm / side streams, two send blocks, the engine's two tuple rings): the pipelined protocol
    (UpdateEndBegin -> gather_tuples_end -> gather_tuples_begin, frame f's tuples travel while frame f + 1 runs) hands out, frame by frame,
    exactly the rows / flags / global env ids of the sequential protocol (Update -> gather_tuples)."""
    import torch
    import deepterrainrl_amd as da_mod
    from deepterrainrl_amd.sharding import ShardedRollout
    pol = dog_policy(om)

    def make(n_local, off):
        return da_mod.BatchScenario("args/opt_args_train_mace.txt", n_local, data_root=REFDATA, extra_args={"terrain_seed": 5, "rand_seed": 3, "global_env_offset": off})
    dev = torch.device("cuda", 0)
    a = ShardedRollout(make, 48, device=dev)
    b = ShardedRollout(make, 48, device=dev, pipelined=True)
    for sr in (a, b):
        sr.broadcast_policy(pol[1], *pol[2:], src=0)
        sr.batch.SetExplore(True, 0.2, 0.025, 0.002)
    frames = 60
    seq = []
    for f in range(frames):
        a.Update()
        a.gather_tuples_begin()
        r, fl, ids = a.gather_tuples_end(dst=0)
        seq.append((r.cpu().numpy().copy(), fl.cpu().numpy().copy(), ids.cpu().numpy().copy()))
    pip = []
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        b.UpdateBegin()
        for f in range(frames):
            if f % 5 == 4:
                b.UpdateEnd(); b.UpdateBegin()
            else:
                b.UpdateEndBegin()
            if b._pending is not None:
                r, fl, ids = b.gather_tuples_end(dst=0)
                pip.append((r.cpu().numpy().copy(), fl.cpu().numpy().copy(), ids.cpu().numpy().copy()))
            b.gather_tuples_begin()
        b.UpdateEnd()
        r, fl, ids = b.gather_tuples_end(dst=0)
        pip.append((r.cpu().numpy().copy(), fl.cpu().numpy().copy(), ids.cpu().numpy().copy()))
    assert len(pip) == frames
    total = 0
    for f in range(frames):
        for x, y in zip(seq[f], pip[f]):
            assert np.array_equal(x, y), f
        total += len(seq[f][0])
    assert total >= 40
    assert b.batch.TupleStats()["dropped"] == 0


TWO_RANK_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import torch, torch.distributed as dist
from conftest import REFDATA, dog_policy
from oracle import model as om
import deepterrainrl_amd as da
from deepterrainrl_amd.sharding import ShardedRollout
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
def make(n, off):
    return da.BatchScenario("args/opt_args_train_mace.txt", n, data_root=REFDATA, extra_args=dict(terrain_seed=300, rand_seed=9, global_env_offset=off))
sr = ShardedRollout(make, {n}, dist=dist, device="cuda:0", pipelined={pipelined}, block_rows=128)   # (a block that takes the lock-step burst of the first cycles whole)
assert sr.staged and sr.world == 2 and "libdtrl.so" in open("/proc/self/maps").read()
pol = dog_policy(om)
if rank == 0:
    sr.broadcast_policy(pol[1], pol[2], pol[3], pol[4], pol[5], src=0)
else:
    sr.broadcast_policy(src=0)
sr.batch.SetExplore(True, 0.2, 0.025, 0.002)
rows, flags, ids = [], [], []
def keep(g):
    if rank == 0:
        rows.append(g[0].cpu().numpy().copy()); flags.append(g[1].cpu().numpy().copy()); ids.append(g[2].cpu().numpy().copy())
if {pipelined}:
    sr.UpdateBegin()
    for f in range({frames} - 1):
        sr.UpdateEndBegin()
        if sr._pending is not None:
            keep(sr.gather_tuples_end(dst=0))
        sr.gather_tuples_begin()
    sr.UpdateEnd()
    keep(sr.gather_tuples_end(dst=0))
    sr.gather_tuples_begin(); keep(sr.gather_tuples_end(dst=0))      # the last frame's ring
else:
    for f in range({frames}):
        sr.Update()
        sr.gather_tuples_begin(); keep(sr.gather_tuples_end(dst=0))
for f in range(4):                       # rows a full block carried over
    sr.gather_tuples_begin(); keep(sr.gather_tuples_end(dst=0))
st = sr.batch.TupleStats()
assert st["pending"] == 0 and st["dropped"] == 0, st
q, qd = sr.batch.PoseVel()
np.savez(os.path.join({out!r}, "rank%d.npz" % rank), q=q, qd=qd, off=sr.offset)
if rank == 0:
    np.savez(os.path.join({out!r}, "tuples.npz"), rows=np.concatenate(rows), flags=np.concatenate(flags), ids=np.concatenate(ids))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("pipelined", [False, True])
def test_two_ranks_sharing_the_gpu_over_gloo(tmp_path, om, pipelined):
    """The N > 1 path on the HIP engine (there is one GPU on the box, and RCCL refuses two ranks on one device): two processes, each with its own libdtrl.so batch on
    cuda:0 and its contiguous range of the 97 global envs (49 + 48), a gloo group whose collectives are staged through pinned host memory (ShardedRollout.staged).
    Rank 0 broadcasts the policy; every frame's tuples are packed on the device and gathered to rank 0. Against ONE process stepping the 97 envs on the same GPU:
    the final pose / velocity of every env is bit-identical (shard-invariant trajectories: terrain and exploration streams are keyed by the global env id), and rank 0
    received every env's tuples, bit for bit and in time order."""
    import subprocess
    import sys
    import deepterrainrl_amd as da_mod
    from conftest import REPO
    n, frames = 97, 60
    script = tmp_path / "worker.py"
    script.write_text(TWO_RANK_WORKER.format(repo=REPO, out=str(tmp_path), n=n, frames=frames, pipelined=bool(pipelined)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(29671 + int(pipelined)), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    pol = dog_policy(om)
    b = da_mod.BatchScenario("args/opt_args_train_mace.txt", n, data_root=REFDATA, extra_args=dict(terrain_seed=300, rand_seed=9))
    b.SetPolicy(pol[1], *pol[2:])
    b.SetExplore(True, 0.2, 0.025, 0.002)
    rows, flags, ids = [], [], []
    for f in range(frames):
        b.Update()
        rr, ff, ii = b.DrainTuples()
        rows.append(rr); flags.append(ff); ids.append(ii)
    rows = np.concatenate(rows); flags = np.concatenate(flags); ids = np.concatenate(ids)
    q, qd = b.PoseVel()
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    assert int(r0["off"]) == 0 and int(r1["off"]) == 49 and len(r0["q"]) == 49 and len(r1["q"]) == 48
    assert np.array_equal(q, np.concatenate([r0["q"], r1["q"]])) and np.array_equal(qd, np.concatenate([r0["qd"], r1["qd"]]))
    t = np.load(tmp_path / "tuples.npz")
    assert len(t["rows"]) == len(rows) and len(rows) >= 100, (len(t["rows"]), len(rows))
    for e in range(n):
        assert np.array_equal(t["rows"][t["ids"] == e], rows[ids == e]) and np.array_equal(t["flags"][t["ids"] == e].astype(np.uint32), flags[ids == e].astype(np.uint32)), e
