"""On-device terrain generation (-terrain_gen= device): the GPU builds and slides every env's cGroundVar2D window at the frame boundary
(deepterrainrl_amd/csrc/dtrl_terrain_dev.h) from a per-env counter stream; the host neither reads a status back nor loops over envs.

What is checked, and against what:
  * the window logic (BuildSegment / InitSegments / Update: C0 seams, flat pad around x = 0, float-rounded origins, segment x ranges) is the SAME template
    code as instantiated on the host in tests/terrain_dev/window_check.cpp with the reference-exact libstdc++ stream -> record-for-record equal to the host
    mode's GroundWindow (which test_reference_pin.py holds bit-exact against the reference's own cGroundVar2D / cTerrainGen2D);
  * with the counter stream: structural invariants of every window, determinism, invariance to sharding and to batch composition, and the generator's
    statistics (gap / step / wall / slope distributions) against the host generator's over thousands of strips;
  * the engine around it: falls -> fresh window + reset, poli_eval distance log, curriculum parameter updates, user resets with seeds.
The GPU twins of these tests are in test_gpu_parity.py (same functions, product library)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REFDATA, REPO, EmulScenario, dog_policy

ARG = "args/dog_slopes_mixed_args.txt"


def make(scn, n, **extra):
    extra.setdefault("terrain_gen", "device")
    return scn(ARG, n, data_root=REFDATA, extra_args=extra)


def set_policy(b, om):
    desc, w, io, isc, oo, osc = dog_policy(om)
    b.SetPolicy(w, io, isc, oo, osc)


def check_window(win, root_x):
    (mn0, mx0, h0), (mn1, mx1, h1) = win
    assert len(h0) >= 2 and len(h1) >= 2 and len(h0) <= 512 and len(h1) <= 512
    assert abs(mx0 - mn1) < 1e-9                                   # the two segments share the seam vertex ...
    # ... at the same height (C0) up to one float rounding: cGroundVar2D::BuildSegment adds h_offset = float(fix - end_h) to every vertex (sim/GroundVar2D.cpp:326-336),
    # and end_h + (fix - end_h) is fix only up to an ulp
    assert abs(float(h0[-1]) - float(h1[0])) <= 2.5e-7 * max(1.0, abs(float(h1[0])))
    assert abs((mx0 - mn0) - (len(h0) - 1) * 0.1) < 1e-6 and abs((mx1 - mn1) - (len(h1) - 1) * 0.1) < 1e-6
    assert mx0 - mn0 >= 20 - 1e-6 and mx1 - mn1 >= 20 - 1e-6        # a strip covers at least the segment width (features overshoot)
    assert mn0 < root_x - 2 + 1e-9 and mx1 > root_x + 11 - 1e-9      # the window covers what the character sees
    assert np.all(np.isfinite(h0)) and np.all(np.isfinite(h1))


def run_device_terrain_rollout(scn, om, n=24, frames=150):
    b = make(scn, n, terrain_seed=5, rand_seed=2)
    set_policy(b, om)
    wins0 = [b.GroundWindow(e) for e in range(n)]
    for e, (win, nb) in enumerate(wins0):
        assert nb == 2
        check_window(win, 0.0)
        # spawn window: [-21, -1] ending at height 0 and [-1, 19] starting at height 0 with one metre of flat padding over x in [0, 1]
        assert -30 < win[0][0] <= -21 + 1e-6 and abs(win[1][0] + 1) < 1e-9 and win[0][2][-1] == 0 and win[1][2][0] == 0   # (strips overshoot by their last feature)
    builds = np.array([w[1] for w in wins0])
    for f in range(frames):
        b.Update()
        if f % 10 == 9:
            q, _ = b.PoseVel()
            for e in range(n):
                win, nb = b.GroundWindow(e)
                check_window(win, q[e, 0])
                assert nb >= builds[e]; builds[e] = nb
    st = b.EvalStats()
    assert st["resets"] >= 3 and st["cycles"] > 5 * n
    assert builds.sum() >= 2 * n + 2 * st["resets"]                   # every reset builds two segments; slides build one
    dist, ids = b.GetDistLog()
    assert len(dist) == st["episodes"]
    return b


def test_device_terrain_rollout_windows_resets_and_dist_log(om):
    run_device_terrain_rollout(EmulScenario, om)


def snapshot(b, envs):
    q, qd = b.PoseVel(envs)
    wins = [b.GroundWindow(int(e)) for e in envs]
    return q, qd, wins


def run_determinism_and_shard_invariance(scn, om):
    """Same global env ids -> same windows and trajectories, whether an env runs in a batch of 12 at offset 0 or of 5 at offset 4, and run after run."""
    frames = 60
    a = make(scn, 12, terrain_seed=9, rand_seed=3); set_policy(a, om)
    a2 = make(scn, 12, terrain_seed=9, rand_seed=3); set_policy(a2, om)
    c = make(scn, 5, terrain_seed=9, rand_seed=3, global_env_offset=4); set_policy(c, om)
    for _ in range(frames):
        a.Update(); a2.Update(); c.Update()
    qa, qda, wa = snapshot(a, np.arange(12)); qb, qdb, wb = snapshot(a2, np.arange(12)); qc, qdc, wc = snapshot(c, np.arange(5))
    assert np.array_equal(qa, qb) and np.array_equal(qda, qdb)
    assert np.array_equal(qa[4:9], qc) and np.array_equal(qda[4:9], qdc)
    for k in range(5):
        (w0, nb0), (w1, nb1) = wa[4 + k], wc[k]
        assert nb0 == nb1
        for s in range(2):
            assert w0[s][0] == w1[s][0] and w0[s][1] == w1[s][1] and np.array_equal(w0[s][2], w1[s][2])
    assert a.EvalStats()["resets"] >= 1
    # RunFrames (everything queued, no host sync between frames) == the same number of Update() calls
    r = make(scn, 12, terrain_seed=9, rand_seed=3); set_policy(r, om)
    r.RunFrames(frames)
    qr, qdr, wr = snapshot(r, np.arange(12))
    assert np.array_equal(qa, qr) and np.array_equal(qda, qdr) and all(np.array_equal(x[0][1][2], y[0][1][2]) for x, y in zip(wa, wr))
    assert r.EvalStats() == a.EvalStats()
    # a different seed gives different ground
    d = make(scn, 2, terrain_seed=10, rand_seed=3)
    assert not np.array_equal(d.GroundWindow(0)[0][1][2][:150], wa[0][0][1][2][:150])


def test_device_terrain_determinism_and_shard_invariance(om):
    run_determinism_and_shard_invariance(EmulScenario, om)


def strip_features(h):
    """(number of vertical jumps > 5 cm, mean |slope| between jumps, fraction of vertices lower than 1 m below the running level = inside gaps)."""
    d = np.diff(h.astype(np.float64))
    jumps = np.abs(d) > 0.05
    smooth = d[~jumps]
    return int(jumps.sum()), float(np.abs(smooth).mean() / 0.1), float((d < -1.0).sum())


def run_generator_statistics(scn, om, terrains=("slopes_mixed", "mixed", "narrow_gaps", "cliffs_rugged")):
    """Device-generated strips vs the host generator's (the reference's algorithm on libstdc++ streams) over many envs: same feature statistics."""
    import deepterrainrl_amd as da_mod
    for name in terrains:
        n = 192
        extra = dict(terrain_seed=77, terrain_file="data/terrain/%s.txt" % name)
        dev = scn("args/dog_slopes_mixed_args.txt", n, data_root=REFDATA, extra_args=dict(extra, terrain_gen="device"))
        host = scn("args/dog_slopes_mixed_args.txt", n, data_root=REFDATA, extra_args=extra)
        fd, fh = [], []
        for e in range(n):
            fd.append(strip_features(dev.GroundWindow(e)[0][1][2][12:]))   # the max segment beyond the flat pad
            fh.append(strip_features(host.GroundWindow(e)[0][1][2][12:]))
        fd, fh = np.array(fd), np.array(fh)
        for k, label in enumerate(("jumps per strip", "mean |slope|", "gap drops per strip")):
            md, mh = fd[:, k].mean(), fh[:, k].mean()
            se = np.sqrt(fd[:, k].var() / n + fh[:, k].var() / n) + 1e-12
            assert abs(md - mh) < 4.5 * se + 1e-9, (name, label, md, mh, se)
        ld = np.array([len(dev.GroundWindow(e)[0][1][2]) for e in range(n)]); lh = np.array([len(host.GroundWindow(e)[0][1][2]) for e in range(n)])
        assert abs(ld.mean() - lh.mean()) < 4.5 * np.sqrt(ld.var() / n + lh.var() / n) + 0.5, (name, ld.mean(), lh.mean())


def test_device_generator_statistics_match_the_host_generator(om):
    run_generator_statistics(EmulScenario, om)


def test_window_logic_equals_the_host_window_on_the_reference_stream():
    """tests/terrain_dev/window_check.cpp: dtrl_terrain_dev.h's window code instantiated with TerrainRand (the reference-exact stream) vs GroundWindow,
    every terrain type, random walks forwards, backwards and out of range."""
    d = os.path.join(REPO, "tests", "terrain_dev")
    subprocess.run(["make", "-s", "-C", d], check=True)
    r = subprocess.run([os.path.join(d, "window_check")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "records compared" in r.stdout and " 0 mismatches" in r.stdout, r.stdout


def run_user_reset_and_curriculum(scn, om):
    b = make(scn, 4, terrain_seed=1, rand_seed=1); set_policy(b, om)
    for _ in range(20):
        b.Update()
    w_before = [b.GroundWindow(e) for e in range(4)]
    b.Reset([1, 3], terrain_seeds=[1234, 99])
    q, _ = b.PoseVel()
    assert abs(q[1, 0]) < 1e-9 and abs(q[3, 0]) < 1e-9 and abs(q[0, 0]) > 0.5
    w_after = [b.GroundWindow(e) for e in range(4)]
    for e in (0, 2):
        assert np.array_equal(w_before[e][0][1][2], w_after[e][0][1][2])
    for e in (1, 3):
        check_window(w_after[e][0], 0.0)
    # the same seed gives the same window again, whatever happened before
    b.Reset([0], terrain_seeds=[1234])
    assert np.array_equal(b.GroundWindow(0)[0][1][2], w_after[1][0][1][2]) and np.array_equal(b.GroundWindow(0)[0][0][2], w_after[1][0][0][2])
    # curriculum: flat parameters (lerp to a parameter set without features is not in this file) -> at least the call reaches the device config
    b.SetTerrainParamsLerp(0.0)
    for _ in range(5):
        b.Update()


def test_device_terrain_user_reset_and_curriculum(om):
    run_user_reset_and_curriculum(EmulScenario, om)
