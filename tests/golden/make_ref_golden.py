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
