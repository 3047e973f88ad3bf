#!/usr/bin/env python3
"""(CPU, needs oracle/_ref/libref_sim.so = /root/reference compiled) How far is Integrator v1 from a Bullet-shaped integrator, behaviourally?

SURVEY 8a row a2 -- what happens inside Bullet's stepSimulation -- is the one part of the hot path that cannot be pinned: Bullet is absent. This tool
quantifies the modelling gap instead of leaving it open: the REFERENCE'S OWN scenario + controllers (compiled unchanged, oracle/_ref/libref_sim.so) are
driven once by Integrator v1 (oracle/or_sim.h = the product kernel's model: reduced coordinates, Delassus-space PGS, Bullet's per-box safe margin and -- since round 5 -- its contact
persistence: warm-started ground contact rows, friction held under an unloaded normal, rows within the breaking threshold; no split impulse; lock-step harness of oracle/refsim.py) and once by oracle/or_bullet_si.h (maximal coordinates, sequential impulse with Bullet 2.8x's published
structure and defaults: ERP joints, 10 sweeps, warm-started persistent contacts, margins + breaking threshold, split impulse, angular-limit rows), on the
same scenes, seeds and (synthetic) policies, and distribution-level statistics of the resulting behaviour are compared:

  cycle_s      gait-cycle duration (time between new-cycle flags)               speed       mean forward speed of the root (m / s)
  falls_k      falls (episode ends) per 1000 env-steps                           duty_front / duty_back   contact-flag duty cycle of the front / back foot
  ep_dist      mean distance per episode (cScenarioPoliEval's dist log)          reward      mean cDogController / cRaptorController::CalcReward at cycle ends

  python tools/a2_deviation.py [--seeds 32] [--frames 300] [--jobs 8] [--null] [--ablate] [--v1-ablate] [--out profiles/r04_a2_deviation.txt]
Every statistic is printed with its standard error over the seeds (each seed an independent run) and every difference in units of its standard error; --null
adds SI on a disjoint seed set (what sampling alone produces). --ablate runs the SI integrator with one Bullet feature removed at a time (margin, warm start, friction
warm start, split impulse, link contacts, 4-point cap, ...): a feature whose removal reproduces Integrator v1 is what carries a v1-vs-SI gap. --v1-ablate runs
Integrator v1 with one modelling switch changed at a time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import model as om  # noqa: E402
from oracle import refsim as rs  # noqa: E402

REF = "/root/reference"
FEET = {"dog": (16, 20), "raptor": (14, 18)}   # front / back end effector: dog finger, toe; raptor right toe, left toe (sim/SimDog.h:11-36, sim/SimRaptor.h)


def policies():
    """synthetic (seeded xavier weights: what rounds 1-4 measured on) or -- A2_POLICY=trained / --policy trained -- the policies tools/learn_curve.py trained through
    the product on the MI355X (tests/golden/policies: Caffe HDF5 + _scale.txt): the regime the reference lives in, a character that crosses the terrain"""
    from conftest import dog_policy
    import test_host_and_emul as T
    if os.environ.get("A2_POLICY", "synthetic") != "trained":
        d = dog_policy(om)
        return {"dog": d, "goat": d, "raptor": T.raptor_policy(om)}      # (the goat scene runs on the dog's net shape: args/goat_cliffs_args.txt names dog_mace3_deploy.prototxt)
    from deepterrainrl_amd import caffe_hdf5
    out = {}
    for name, net, stem in (("dog", "data/policies/dog/nets/dog_mace3_deploy.prototxt", "dog_mace3_slopes_mixed_model"),
                            ("goat", "data/policies/dog/nets/dog_mace3_deploy.prototxt", "goat_mace3_cliffs_model"),
                            ("raptor", "data/policies/raptor/nets/raptor_mace3_deploy.prototxt", "raptor_mace3_narrow_gaps_model")):
        desc = om.parse_deploy_prototxt(os.path.join(REF, net))
        base = os.path.join(REPO, "tests", "golden", "policies", stem)
        w = caffe_hdf5.load_mace_weights(base + ".h5", desc.n_frags)
        out[name] = (desc, w) + tuple(om.load_scale_file(base + "_scale.txt"))
    return out


SCENES = [  # tag, arg file, character, policy?, dims
    ("dog flat FSM", "args/sim_dog_args.txt", "dog", None),
    ("raptor flat FSM", "args/sim_raptor_args.txt", "raptor", None),
    ("dog slopes_mixed + MACE net (configs[1])", "args/dog_slopes_mixed_args.txt", "dog", "dog"),
    ("raptor narrow_gaps + MACE net (configs[2])", "args/raptor_narrow_gaps_args.txt", "raptor", "raptor"),
    ("goat cliffs_rugged + MACE net (configs[4] scene)", "args/goat_cliffs_args.txt", "dog", "goat"),
]


def raw_forward(e, pol):
    desc, w, io, isc, oo, osc = pol

    def raw(x_norm):
        x_raw = np.where(isc != 0, x_norm / np.where(isc != 0, isc, 1.0), 0.0) - io
        y = e.nn_eval(x_raw)
        return (y + oo) * osc
    return raw


class Stats:
    def __init__(self, feet):
        self.feet = feet
        self.steps = 0; self.cycles = []; self.speed = 0.0; self.duty = np.zeros(2); self.rewards = []; self.falls = 0; self.dists = []
        self._last_cycle_step = None

    def env_step(self, contacts, flags, vx, dt=1.0 / 600.0):
        self.steps += 1; self.speed += vx
        self.duty += [contacts[self.feet[0]] != 0, contacts[self.feet[1]] != 0]
        if flags & 4:
            if self._last_cycle_step is not None:
                self.cycles.append((self.steps - self._last_cycle_step) * dt)
            self._last_cycle_step = self.steps

    def episode_end(self):
        self._last_cycle_step = None

    def raw(self):
        return dict(steps=self.steps, cycles=list(self.cycles), speed=self.speed, duty=self.duty.tolist(), rewards=[float(x) for x in self.rewards], falls=self.falls,
                    dists=[float(x) for x in self.dists])

    @classmethod
    def merged(cls, raws):
        st = cls((0, 0))
        for r in raws:
            st.steps += r["steps"]; st.cycles += r["cycles"]; st.speed += r["speed"]; st.duty += np.array(r["duty"]); st.rewards += r["rewards"]; st.falls += r["falls"]
            st.dists += r["dists"]
        return st

    def summary(self):
        n = max(self.steps, 1)
        c = np.array(self.cycles) if self.cycles else np.zeros(1)
        return dict(env_steps=self.steps, cycle_s=float(c.mean()), cycle_median=float(np.median(c)), cycle_long=float((c > 0.6).mean()), cycle_sd=float(c.std()), n_cycles=len(self.cycles), speed=self.speed / n, falls_k=1000.0 * self.falls / n,
                    duty_front=float(self.duty[0] / n), duty_back=float(self.duty[1] / n), ep_dist=float(np.mean(self.dists)) if self.dists else float("nan"),
                    n_episodes=len(self.dists), reward=float(np.nanmean(self.rewards)) if self.rewards else float("nan"))


_POLS = None


def run_seed(job):
    """One seed of one cell (a worker process of the pool): returns the raw accumulators."""
    global _POLS
    scene_idx, integrator, seed, frames, si_opts, v1_overrides = job
    if _POLS is None:
        _POLS = policies()
    pols = _POLS
    tag, arg, char, polname = SCENES[scene_idx]
    st = Stats(FEET[char])
    m, _ = om.build_model(arg, REF, overrides=v1_overrides or {})
    pol = pols[polname] if polname else None
    e = om.OracleEnv(m, terrain_seed=seed, policy=pol)
    if pol is not None:
        rs.nn_config(e.S if hasattr(e, "S") else len(pol[2]), len(pol[4]), raw_forward(e, pol))
    r = rs.RefScenario("poli_eval", arg, REF, global_seed=seed + 1)
    if pol is not None:
        r.set_net_scale(*pol[2:])
    r.seed_ground_and_reset(seed)
    n_log = 0
    if integrator == "v1":
        ls = rs.LockStep(r, e)
        for f in range(frames):
            k0 = len(ls.records)
            ls.update(); e.frame_end()
            for o, rr in ls.records[k0:]:
                if rr.get("after_reset"):
                    continue
                st.env_step(rr["contacts"], rr["flags"], rr["qd"][0])
                if rr["flags"] & 4:
                    st.rewards.append(r.calc_reward())
            log = r.eval_stats()["dist_log"]
            if len(log) > n_log:
                st.dists += list(log[n_log:]); n_log = len(log)
            if ls.records[-1][1].get("after_reset"):
                st.falls += 1; st.episode_end()
            del ls.records[:]
    else:
        r.use_bullet_si(**(si_opts or {}))

        seen = [False]

        def observe(dt=0, n=0):
            # called in front of every env-step's physics (and once after Update): the state the previous env-step left behind, as in LockStep's records
            if not seen[0]:
                seen[0] = True; return
            fl = r.flags()
            st.env_step(r.contact_flags(), fl, r.pose_vel()[1][0])
            if fl & 4:
                st.rewards.append(r.calc_reward())
        r.set_step_hook(observe)
        for f in range(frames):
            t0 = r.time()
            r.update()
            fell = r.time() < t0 + 0.5 / 30.0      # the scenario reset inside Update (fall): its last env-step is gone
            if not fell:
                observe()
            seen[0] = False
            log = r.eval_stats()["dist_log"]
            if len(log) > n_log:
                st.dists += list(log[n_log:]); n_log = len(log)
            if fell:
                st.falls += 1; st.episode_end()
    return st.raw()


KEYS = ("cycle_s", "cycle_median", "cycle_long", "speed", "falls_k", "duty_front", "duty_back", "ep_dist", "reward")
_POOL = None


def run(scene, integrator, seeds, frames, pols=None, si_opts=None, v1_overrides=None, jobs=1):
    """All seeds of one cell. Returns the pooled summary plus `se`: the standard error of every statistic over the seeds (each seed an independent 300-frame
    run; the per-seed value of a ratio statistic is that seed's own ratio) -- what the v1-vs-SI differences have to be read against."""
    global _POOL
    idx = SCENES.index(scene)
    work = [(idx, integrator, sd, frames, si_opts, v1_overrides) for sd in seeds]
    if jobs > 1:
        if _POOL is None:
            import multiprocessing as mp
            _POOL = mp.get_context("fork").Pool(jobs)
        raws = _POOL.map(run_seed, work, chunksize=1)
    else:
        raws = [run_seed(w) for w in work]
    out = Stats.merged(raws).summary()
    per = [Stats.merged([r]).summary() for r in raws]
    out["se"] = {}
    for k in KEYS:
        v = np.array([p[k] for p in per], float); v = v[np.isfinite(v)]
        out["se"][k] = float(v.std(ddof=1) / np.sqrt(len(v))) if len(v) > 1 else float("nan")
    return out


def fmt(s):
    return "  ".join("%s %8.4f (%.4f)" % (k, s[k], s["se"][k]) for k in KEYS) + "   (cycles %d, episodes %d, env-steps %d)" % (s["n_cycles"], s["n_episodes"], s["env_steps"])


def rel_line(a, b):
    """(a - b) / b per statistic, and z = the difference in units of its standard error over seeds"""
    parts = []
    for k in KEYS:
        if b[k] and np.isfinite(b[k]) and np.isfinite(a[k]):
            se = np.hypot(a["se"][k], b["se"][k])
            parts.append("%s %+6.1f%% (%+.1f s.e.)" % (k, 100.0 * (a[k] - b[k]) / b[k], (a[k] - b[k]) / se if se > 0 else float("nan")))
        else:
            parts.append("%s     n/a" % k)
    return "  ".join(parts)


SI_ABLATIONS = (("no margin", dict(use_margin=0)), ("no warm start", dict(warmstarting=0)), ("no split impulse", dict(split_impulse=0)),
                ("no link contacts", dict(link_contacts=0)), ("1 point per pair", dict(max_points=1)), ("erp 0.8 joints", dict(erp=0.8)), ("20 iterations", dict(iterations=20)),
                ("no FRICTION warm start (normals only)", dict(friction_warmstart=0)), ("no terrain-vertex contacts", dict(vertex_contacts=0)),
                ("round-3 comparator: margin 0.04 + breaking 0.02 on every box", dict(safe_margin=0, relative_breaking=0)),
                ("diagnostic: friction rows always resolved (no `if (totalImpulse > 0)`)", dict(friction_skip=0)), ("diagnostic: friction along the plane-space vector only", dict(friction_dir=0)),
                ("diagnostic: normal and friction row of a contact interleaved", dict(interleave=1)), ("diagnostic: friction warm start on GROUND contacts only", dict(friction_ws_lifted=3)),
                ("diagnostic: friction warm start on LINK--LINK contacts only", dict(friction_ws_lifted=4)))
V1_ABLATIONS = (("round-4 model: no warm start, rows only while penetrating (-warm_start= 0 -contact_breaking= 0)", dict(warm_start=0, contact_breaking=0)),
                ("no warm start / friction rule (-warm_start= 0), rows within the breaking threshold", dict(warm_start=0)),
                ("Bullet's rule, rows only while penetrating (-contact_breaking= 0)", dict(contact_breaking=0)),
                ("Bullet's rule with the interleaved sweep (oracle-only -warm_start= 2)", dict(warm_start=2)),
                ("Bullet's rule with Bullet's friction direction: along the pre-solve tangential velocity (oracle-only -warm_start= 3)", dict(warm_start=3)),
                ("plain warm start of every row, no friction rule (oracle-only -warm_start= 4)", dict(warm_start=4)),
                ("sharp boxes (-collision_margin= 0)", dict(collision_margin=0)), ("no link contacts", dict(link_contacts=0)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=6)
    ap.add_argument("--seed0", type=int, default=101)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default="")
    ap.add_argument("--ablate", action="store_true", help="SI with one Bullet feature removed at a time")
    ap.add_argument("--v1-ablate", action="store_true", help="Integrator v1 with one modelling switch changed at a time")
    ap.add_argument("--null", action="store_true", help="also run SI on a DISJOINT seed set: the SI-vs-SI line is what sampling alone produces")
    ap.add_argument("--scenes", default="", help="comma-separated scene indices (default: all)")
    ap.add_argument("--policy", choices=["synthetic", "trained"], default="synthetic", help="trained: tests/golden/policies (scenes 2 and 3 are the ones they were trained for)")
    a = ap.parse_args()
    assert rs.available(), "oracle/_ref/libref_sim.so missing: make -C oracle/_ref_build"
    os.environ["A2_POLICY"] = a.policy       # (read by the pool's workers)
    seeds = list(range(a.seed0, a.seed0 + a.seeds))
    scenes = [SCENES[int(i)] for i in a.scenes.split(",")] if a.scenes else SCENES
    lines = ["# Integrator v1 (product model) vs Bullet-shaped sequential impulse (oracle/or_bullet_si.h), both driven by the REFERENCE'S OWN controllers",
             "# %d seeds x %d outer frames (%d env-steps) per cell; value (standard error over seeds); rel = (v1 - SI) / SI with the difference in standard errors; tools/a2_deviation.py"
             % (a.seeds, a.frames, a.seeds * a.frames * 20)]
    if a.policy == "trained":
        lines.append("# POLICIES: trained through the product (tools/learn_curve.py, 60 000 iterations; tests/golden/policies), not the seeded xavier weights of the other studies")
    results = {}
    for sc in scenes:
        t0 = time.time(); n0 = len(lines)
        v1 = run(sc, "v1", seeds, a.frames, jobs=a.jobs)
        si = run(sc, "si", seeds, a.frames, jobs=a.jobs)
        results[sc[0]] = {"v1": v1, "si": si}
        lines.append("\n## %s" % sc[0])
        lines.append("  v1   " + fmt(v1))
        lines.append("  SI   " + fmt(si))
        lines.append("  rel  " + rel_line(v1, si))
        if a.null:
            si2 = run(sc, "si", [sd + 1000 for sd in seeds], a.frames, jobs=a.jobs)
            results[sc[0]]["si, disjoint seeds"] = si2
            lines.append("  SI, disjoint seeds   " + fmt(si2))
            lines.append("  null (SI' - SI) / SI " + rel_line(si2, si))
        if a.v1_ablate:
            for name, ov in V1_ABLATIONS:
                ab = run(sc, "v1", seeds, a.frames, v1_overrides=ov, jobs=a.jobs)
                results[sc[0]]["v1, " + name] = ab
                lines.append("  v1, %s\n       %s\n       vs SI: %s" % (name, fmt(ab), rel_line(ab, si)))
        if a.ablate:
            for name, opts in SI_ABLATIONS:
                ab = run(sc, "si", seeds, a.frames, si_opts=opts, jobs=a.jobs)
                results[sc[0]]["si, " + name] = ab
                lines.append("  SI, %s\n       %s\n       v1 vs this: %s" % (name, fmt(ab), rel_line(v1, ab)))
        print("\n".join(lines[n0:]), "[%.0f s]" % (time.time() - t0), flush=True)
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
        json.dump(results, open(os.path.splitext(a.out)[0] + ".json", "w"), indent=1)
    return results


if __name__ == "__main__":
    main()
