#!/usr/bin/env python3
"""Freeze what the REFERENCE'S OWN scenario / controller / ground code (oracle/_ref/libref_sim.so = the sources under /root/reference compiled unchanged by
oracle/_ref_build/Makefile) computes in the lock-step runs of tests/test_reference_sim.py on the scenes of the BASELINE configurations, so that the PRODUCT
can be stepped against the reference on a box without /root/reference (the MI355X box): tests/golden/ref_golden_configs.npz

  <run>/frame/{ws_n, ws_id, ws_lam}    the integrator's persistent contact rows at the end of every outer frame (identities + impulses: dtrl_set_contact_cache)
  <run>/frame/{q, qd}                  the motion both sides rode (oracle/or_sim.h supplies it: the inside of Bullet's stepSimulation is the one part of
                                       the reference that cannot be compiled here), at the end of every outer frame, after the frame's fall / reset logic
  <run>/frame/{tau, contacts, state, phase, action_id, params, pd_targets, flags, after_reset, cycles, episodes, avg_dist}
                                       the REFERENCE's view at the end of every outer frame: clamped joint torques (cJoint), contact flags (cContactManager),
                                       FSM state / phase, current action id and parameters, PD targets, flag word, cScenarioPoliEval counters
  <run>/cycle/{frame, poli_state}      the REFERENCE's RecordPoliState at every frame in which a new cycle began
  <run>/dist_log                       the REFERENCE's cScenarioPoliEval::GetDistLog at the end of the run
  <run>/tuples/{rows, flags, frame}    (exp runs) the tuples the REFERENCE's cScenarioExp(MACE) recorded, [r | s | a | s'] rows, flag words, frame of arrival
  <run>/step/{tau, contacts, state, phase}   the first 240 env-steps at env-step resolution (no fall happens that early)

Runs: dog_sm32 / dog_sm9 (configs[1]: dog + slopes_mixed + MACE net, poli_eval, through falls and resets), raptor_ng (configs[2]: raptor + narrow_gaps,
stance-mirrored state), goat_cliffs (configs[4]'s scene: goat + cliffs_rugged, one substep per env-step), exp_mace / exp_q / raptor_exp_mace (tuples);
round 6: dog_sm_trained / raptor_ng_trained / goat_trained = the three scenes under the policies trained through the engine (tests/golden/policies).
Run from the repo root in the container that has /root/reference:  python tests/golden/make_ref_golden_configs.py
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import model as om  # noqa: E402
from oracle import refsim as rs  # noqa: E402
from conftest import dog_policy, trained_policy  # noqa: E402
import test_host_and_emul as T  # noqa: E402

REF = "/root/reference"


def freeze(out, tag, kind, arg, seed, frames, pol, S, O, global_seed, explore_off=False, command=None, overrides=None, q_head=None, stop_after_reset=False):
    m, _ = om.build_model(arg, REF, overrides=overrides or {})
    if explore_off:
        m.enable_explore = 0
    e = om.OracleEnv(m, terrain_seed=seed, policy=pol)
    if q_head:
        fwd9 = T_policy_raw(e, pol)
        rs.nn_config(S, O, lambda x: fwd9(x)[1:1 + O])
    else:
        rs.nn_config(S, O, T_policy_raw(e, pol))
    r = rs.RefScenario(kind, arg, REF, global_seed=global_seed)
    if q_head:
        r.set_net_scale(pol[2], pol[3], q_head[0], q_head[1])
    else:
        r.set_net_scale(*pol[2:])
    r.seed_ground_and_reset(seed)
    if explore_off:
        r.enable_explore(0)
    if command is not None:
        r.command_action(command); e.command_action(command)
    ls = rs.LockStep(r, e)
    F = {k: [] for k in ("q", "qd", "tau", "contacts", "state", "phase", "action_id", "params", "pd_targets", "flags", "after_reset", "cycles", "episodes", "avg_dist", "ws_n", "ws_id", "ws_lam")}
    cyc_f, cyc_s = [], []
    t_rows, t_flags, t_frame = [], [], []
    prev_cycles = 0
    for f in range(frames):
        ls.update(); e.frame_end()
        q, qd = e.pose_vel()
        _, rr = ls.records[-1]
        F["q"].append(q); F["qd"].append(qd)
        wn, wid, wlam = e.warm_cache(); F["ws_n"].append(wn); F["ws_id"].append(wid.copy()); F["ws_lam"].append(wlam.copy())   # the persistent contact rows: with (q, qd) the whole dynamic state
        F["tau"].append(rr["tau"]); F["contacts"].append(rr["contacts"].astype(np.int8)); F["state"].append(rr["state"]); F["phase"].append(rr["phase"])
        F["action_id"].append(rr["action_id"]); F["params"].append(rr["params"]); F["pd_targets"].append(rr["pd_targets"]); F["flags"].append(rr["flags"])
        F["after_reset"].append(bool(rr.get("after_reset")))
        if kind == "poli_eval":
            st = r.eval_stats()
            F["cycles"].append(st["cycles"]); F["episodes"].append(st["episodes"]); F["avg_dist"].append(st["avg_dist"])
            if st["cycles"] != prev_cycles:
                prev_cycles = st["cycles"]
                cyc_f.append(f); cyc_s.append(r.poli_state())
        else:
            a, fa = r.drain_tuples()
            b, fb = e.drain_tuples(f64=True)
            assert len(a) == len(b)
            for x, p in zip(a, fa):
                t_rows.append(x); t_flags.append(p); t_frame.append(f)
            if stop_after_reset and e.stats()["resets"] > 0 and len(t_rows) >= 4:
                break
    for k, v in F.items():
        if v:
            out["%s/frame/%s" % (tag, k)] = np.array(v)
    if kind == "poli_eval":
        out["%s/cycle/frame" % tag] = np.array(cyc_f, np.int32); out["%s/cycle/poli_state" % tag] = np.array(cyc_s)
        out["%s/dist_log" % tag] = r.eval_stats()["dist_log"]
    else:
        out["%s/tuples/rows" % tag] = np.array(t_rows); out["%s/tuples/flags" % tag] = np.array(t_flags, np.uint32); out["%s/tuples/frame" % tag] = np.array(t_frame, np.int32)
    recs = [rr for _, rr in ls.records[:240]]
    out["%s/step/tau" % tag] = np.array([x["tau"] for x in recs]); out["%s/step/contacts" % tag] = np.array([x["contacts"] for x in recs], np.int8)
    out["%s/step/state" % tag] = np.array([x["state"] for x in recs], np.int32); out["%s/step/phase" % tag] = np.array([x["phase"] for x in recs])
    print(tag, "frames", len(F["q"]), "cycles", len(cyc_f), "tuples", len(t_rows), "resets", e.stats()["resets"], "episodes", e.stats().get("episodes"))


def T_policy_raw(e, pol):
    import test_reference_sim as TR
    return TR._policy_raw_forward(e, pol)


def main():
    assert rs.available(), "oracle/_ref/libref_sim.so missing: make -C oracle/_ref_build"
    out = {}
    dog = dog_policy(om)
    rap = T.raptor_policy(om)
    freeze(out, "dog_sm32", "poli_eval", "args/dog_slopes_mixed_args.txt", 32, 150, dog, 283, 90, 9)     # falls at frames 36 and 73
    freeze(out, "dog_sm9", "poli_eval", "args/dog_slopes_mixed_args.txt", 9, 150, dog, 283, 90, 9)       # falls at frames 85 and 119
    freeze(out, "raptor_ng", "poli_eval", "args/raptor_narrow_gaps_args.txt", 11, 90, rap, 275, 87, 2)
    freeze(out, "goat_cliffs", "poli_eval", "args/goat_cliffs_args.txt", 8, 120, dog, 283, 90, 3)
    # round 6: the same scenes under the policies trained THROUGH the engine (tests/golden/policies): long contact-rich episodes, the regime the reference lives in
    freeze(out, "dog_sm_trained", "poli_eval", "args/dog_slopes_mixed_args.txt", 41, 150, trained_policy(om, "dog"), 283, 90, 9)
    freeze(out, "raptor_ng_trained", "poli_eval", "args/raptor_narrow_gaps_args.txt", 42, 150, trained_policy(om, "raptor"), 275, 87, 2)
    freeze(out, "goat_trained", "poli_eval", "args/goat_cliffs_args.txt", 43, 150, trained_policy(om, "goat"), 283, 90, 3)
    freeze(out, "exp_mace", "exp_mace", "args/opt_args_train_mace.txt", 21, 120, dog, 283, 90, 4, explore_off=True, command=2, overrides={"policy_model": ""}, stop_after_reset=True)
    freeze(out, "raptor_exp_mace", "exp_mace", "args/opt_args_train_raptor_mace.txt", 28, 150, rap, 275, 87, 7, explore_off=True, command=0, overrides={"policy_model": ""}, stop_after_reset=True)
    freeze(out, "raptor_exp_mace29", "exp_mace", "args/opt_args_train_raptor_mace.txt", 29, 150, rap, 275, 87, 7, explore_off=True, command=0, overrides={"policy_model": ""}, stop_after_reset=True)   # rounds 3-4's seed, kept beside 28
    # Q head: the oracle holds the single-head net in the padded MACE form (one unused critic slot in front)
    desc = om.parse_deploy_prototxt(os.path.join(REF, "data/policies/dog/nets/dog_q_deploy.prototxt"))
    w = om.actor_xavier_weights(desc, 5)
    io, isc = np.zeros(283), np.ones(283)
    oo, osc = -0.5 * np.ones(8), 2 * np.ones(8)
    wm, oom, osm = om.actor_policy_to_mace(desc, w, oo, osc)
    freeze(out, "exp_q", "exp", "args/opt_args_train_q.txt", 33, 150, (desc, wm, io, isc, oom, osm), 283, 8, 6, explore_off=True, command=1, q_head=(oo, osc), stop_after_reset=True)
    # compact storage: torques / targets / parameters / policy states as float32 (compared at >= 1e-5 relative), the motion in float64
    for k in list(out):
        if out[k].dtype == np.float64 and not (k.endswith("/frame/q") or k.endswith("/frame/qd") or k.endswith("/dist_log") or k.endswith("/avg_dist") or k.endswith("/phase")):
            out[k] = out[k].astype(np.float32)
    path = os.path.join(REPO, "tests", "golden", "ref_golden_configs.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d arrays, %d bytes" % (path, len(out), os.path.getsize(path)))


if __name__ == "__main__":
    main()
