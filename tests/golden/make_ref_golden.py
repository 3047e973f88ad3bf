#!/usr/bin/env python3
"""Freeze outputs of the REFERENCE'S OWN code (oracle/_ref/libref_core.so, built by oracle/_ref_build/Makefile from the sources under
/root/reference) into tests/golden/ref_golden.npz, so that the pin holds on boxes without the reference checkout or the library:
  terrain/<type>/<param set>/<seed>/<width>   float32 strips of cTerrainGen2D's 14 terrain functions
  params/<param set>                          the 40-vectors used
  rbd/<char>/{q, qd, H, C, grav, body_pos}    cRBDModel::Update + BuildMassMat / BuildBiasForce / CalcGravityForce, cKinTree::CalcBodyPartPos
Run from the repo root in the container that has /root/reference:  python tests/golden/make_ref_golden.py
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import model as om  # noqa: E402  (only for reading the arg / data files)
from oracle import refcore as rc  # noqa: E402

REF = "/root/reference"
TYPES = ["flat", "gaps", "steps", "walls", "bumps", "mixed", "narrow_gaps", "slopes", "slopes_gaps", "slopes_steps", "slopes_walls",
         "slopes_mixed", "slopes_narrow_gaps", "cliffs"]
out = {}
sets = {"default": rc.terrain_default_params()}
for tag, f in (("slopes_mixed", "slopes_mixed.txt"), ("cliffs_rugged", "cliffs_rugged.txt")):
    _, ps = rc.terrain_load_file(os.path.join(REF, "data", "terrain", f))
    sets[tag] = ps[0]
for k, v in sets.items():
    out["params/" + k] = v
for t in TYPES:
    for pset, seed, width in (("default", 3, 20.0), ("slopes_mixed" if t != "cliffs" else "cliffs_rugged", 20260925, 20.0)):
        h, _ = rc.terrain_build(t, sets[pset], seed, width)
        out["terrain/%s/%s/%d/%g" % (t, pset, seed, width)] = h
rng = np.random.RandomState(2026)
for name, arg in (("dog", "args/dog_slopes_mixed_args.txt"), ("goat", "args/goat_cliffs_args.txt"), ("raptor", "args/raptor_narrow_gaps_args.txt")):
    m, info = om.build_model(arg, REF)
    r = rc.RefChar(os.path.join(REF, info["args"]["character_file"]))
    D = r.D
    q0 = np.array(m.pose0[:D]); qd0 = np.array(m.vel0[:D])
    Q = [q0] + [q0 + rng.uniform(-0.7, 0.7, D) for _ in range(5)]
    QD = [qd0] + [qd0 + rng.uniform(-4, 4, D) for _ in range(5)]
    res = [r.rbd(q, qd) for q, qd in zip(Q, QD)]
    out["rbd/%s/q" % name] = np.array(Q); out["rbd/%s/qd" % name] = np.array(QD)
    for k in ("H", "C", "grav"):
        out["rbd/%s/%s" % (name, k)] = np.array([x[k] for x in res])
    out["rbd/%s/body_pos" % name] = np.array([r.kin_bodies(q)[0] for q in Q])
path = os.path.join(REPO, "tests", "golden", "ref_golden.npz")
np.savez_compressed(path, **out)
print("wrote %s: %d arrays, %d bytes" % (path, len(out), os.path.getsize(path)))

# ---- lock-step traces: what the REFERENCE'S controller / contact manager / torque clamp (oracle/_ref/libref_sim.so) computed while the oracle supplied
# the motion (oracle/refsim.py LockStep), env-step by env-step, for the two FSM scenes of BASELINE configs[0]. The product (HIP on the GPU box, lane-loop
# build here) is checked against these arrays: lockstep/<char>/{q, tau, contacts, state, phase, pd_targets}
from oracle import refsim as rs  # noqa: E402
if rs.available():
    for name, arg, frames in (("dog", "args/sim_dog_args.txt", 12), ("raptor", "args/sim_raptor_args.txt", 12)):
        m, _ = om.build_model(arg, REF)
        e = om.OracleEnv(m, terrain_seed=5)
        r = rs.RefScenario("sim_char", arg, REF, global_seed=3)
        r.seed_ground_and_reset(5)
        ls = rs.LockStep(r, e)
        for f in range(frames):
            ls.update()
        recs = ls.records
        out["lockstep/%s/q" % name] = np.array([o["q"] for o, _ in recs])
        out["lockstep/%s/qd" % name] = np.array([o["qd"] for o, _ in recs])
        out["lockstep/%s/tau" % name] = np.array([rr["tau"] for _, rr in recs])                    # the reference's clamped joint torques (joint j at index j)
        out["lockstep/%s/contacts" % name] = np.array([rr["contacts"] for _, rr in recs], np.int8)
        out["lockstep/%s/state" % name] = np.array([rr["state"] for _, rr in recs], np.int32)
        out["lockstep/%s/phase" % name] = np.array([rr["phase"] for _, rr in recs])
        out["lockstep/%s/pd_targets" % name] = np.array([rr["pd_targets"] for _, rr in recs])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_golden.npz"), **out)
    print("lock-step traces added:", [k for k in out if k.startswith("lockstep/")][:4], "...")
