#!/usr/bin/env python3
"""(CPU, needs /root/reference) What a LONG gait cycle of configs[2] is, on both integrators -- VERDICT r5 #1c.
The raptor's step ends when the swing toe touches (sim/RaptorController.cpp:11-37: Contact / Down / Passing advance on time, Up on swing-toe contact), so a cycle longer
than 0.6 s is an Up state that lasts > 0.35 s. For every cycle of every seed this records, on Integrator v1 (lock-step harness) and on the Bullet-shaped comparator
(oracle/or_bullet_si.h, Bullet's defaults): the dwell in Up, and over the Up dwell the root height / pitch, whether the STANCE toe is in contact, which torso links
touch the ground, the forward speed, the swing toe's height above the ground, and how the cycle ended (swing-toe contact, or the episode's fall).
  python tools/a2_long_cycles.py --seeds 16 --out profiles/r06_a2_long_cycles.txt"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import a2_deviation as a2  # noqa: E402
from a2_deviation import om, rs, REF  # noqa: E402

R_TOE, L_TOE = 14, 18           # sim/SimRaptor.h joint enum: right_toe, left_toe
TORSO = list(range(0, 11))       # root, spine0-3, head, tail0-4


def run_seed(job):
    integrator, seed, frames, si_opts, v1_overrides = job
    pols = a2.policies()
    tag, arg, char, polname = a2.SCENES[3]
    m, _ = om.build_model(arg, REF, overrides=v1_overrides or {})
    pol = pols[polname]
    e = om.OracleEnv(m, terrain_seed=seed, policy=pol)
    rs.nn_config(len(pol[2]), len(pol[4]), a2.raw_forward(e, pol))
    r = rs.RefScenario("poli_eval", arg, REF, global_seed=seed + 1)
    r.set_net_scale(*pol[2:])
    r.seed_ground_and_reset(seed)
    steps = []      # per env-step: (state, new_cycle, stance_toe_contact, swing_toe_contact, n_torso_contacts, root_y_above_ground, pitch, vx, swing_toe_clearance)

    def snap():
        st, ph, aid, prm, tg, act = r.ctrl()
        con = r.contact_flags(); fl = r.flags()
        q, qd = r.pose_vel()
        p, a_, v, w = r.bodies()
        # stance: the controller mirrors the state by stance; the swing toe is the one whose contact ends the step. Not exposed: infer both toes
        gy = r.sample_ground(q[0])[0]
        clr = [p[t][1] - r.sample_ground(p[t][0])[0] for t in (R_TOE, L_TOE)]
        slip = [abs(v[t][0]) if con[t] else np.nan for t in (R_TOE, L_TOE)]     # tangential speed of a toe that is flagged in contact (world x: the terrain is near level between gaps)
        steps.append((st, 1 if fl & 4 else 0, int(con[R_TOE]), int(con[L_TOE]), int(sum(con[j] for j in TORSO)), q[1] - gy, q[2], qd[0], clr[0], clr[1], np.nanmin(slip) if np.isfinite(slip).any() else np.nan, r.com()[1][1]))
    falls = []
    if integrator == "v1":
        ls = rs.LockStep(r, e)
        orig = ls._hook
        def hook(dt, n):
            if ls.snap is not None:
                snap()
            orig(dt, n)
        r.set_step_hook(hook)
        for f in range(frames):
            ls.update(); e.frame_end()
            if ls.records[-1][1].get("after_reset"):
                falls.append(len(steps))
            else:
                snap()
            del ls.records[:]
    else:
        r.use_bullet_si(**(si_opts or {}))
        seen = [False]
        def observe(dt=0, n=0):
            if not seen[0]:
                seen[0] = True; return
            snap()
        r.set_step_hook(observe)
        for f in range(frames):
            t0 = r.time(); r.update()
            fell = r.time() < t0 + 0.5 / 30.0
            if not fell:
                observe()
            else:
                falls.append(len(steps))
            seen[0] = False
    return np.array(steps, float), falls


def cycles_of(steps, falls):
    """list of dicts per cycle: duration, per-state dwell, Up-state statistics, how it ended"""
    out = []
    starts = [i for i in range(len(steps)) if steps[i][1]]
    ends = sorted(set(starts[1:] + falls + [len(steps)]))
    fallset = set(falls)
    for s in starts:
        e_ = min(x for x in ends if x > s)
        seg = steps[s:e_]
        up = seg[seg[:, 0] == 3]
        d = dict(dur=len(seg) / 600.0, up=len(up) / 600.0, ended="fall" if e_ in fallset else ("toe" if e_ in starts else "run end"))
        on = (seg[:, 2] + seg[:, 3]) > 0
        if on.any():
            d["slip"] = float(np.nanmean(seg[on, 10])); d["stance"] = float(on.mean() * len(seg) / 600.0)
            last = np.nonzero(on)[0][-1]
            d["vy_off"] = float(seg[last, 11])                       # COM vertical velocity at the last env-step of the cycle with a toe on the ground (take-off)
        if len(up):
            d.update(root_h=up[:, 5].mean(), pitch=up[:, 6].mean(), torso=float((up[:, 4] > 0).mean()), vx=up[:, 7].mean(), toes_down=float(((up[:, 2] + up[:, 3]) > 0).mean()),
                     clr_min=float(np.minimum(up[:, 8], up[:, 9]).mean()), clr_max=float(np.maximum(up[:, 8], up[:, 9]).mean()))
        out.append(d)
    return out


def describe(tag, cyc):
    lines = []
    dur = np.array([c["dur"] for c in cyc])
    for name, sel in (("all", np.ones(len(cyc), bool)), ("<= 0.6 s", dur <= 0.6), ("> 0.6 s", dur > 0.6)):
        cs = [c for c, k in zip(cyc, sel) if k and "root_h" in c]
        if not cs:
            lines.append("  %-9s none" % name); continue
        g = lambda k: np.mean([c[k] for c in cs])
        ended = {k: sum(1 for c in cs if c["ended"] == k) for k in ("toe", "fall", "run end")}
        sl = [c["slip"] for c in cs if "slip" in c]; vo = np.array([c["vy_off"] for c in cs if "vy_off" in c])
        extra = "" if not sl else " | toe slip speed while in contact %.3f m/s, stance %.3f s, COM vy at take-off: mean %.2f, 90 %% %.2f, share > 2 m/s %.3f" % (np.mean(sl), np.mean([c["stance"] for c in cs if "stance" in c]), vo.mean(), np.quantile(vo, 0.9), (vo > 2).mean())
        lines.append("  %-9s n %4d (%.3f of cycles)  mean %.3f s, in Up %.3f s | over Up: root height %.3f m, pitch %+.2f rad, a torso link on the ground %.2f of the time, a toe on the ground %.2f, "
                     "lower / higher toe clearance %.3f / %.3f m, vx %.2f m/s | ended by swing-toe contact %d, fall %d, end of run %d" %
                     (name, len(cs), len(cs) / max(len(cyc), 1), g("dur"), g("up"), g("root_h"), g("pitch"), g("torso"), g("toes_down"), g("clr_min"), g("clr_max"), g("vx"), ended["toe"], ended["fall"], ended["run end"]) + extra)
    return ["%s: %d cycles" % (tag, len(cyc))] + lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=16); ap.add_argument("--seed0", type=int, default=101); ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--jobs", type=int, default=os.cpu_count() or 1); ap.add_argument("--out", default="")
    a = ap.parse_args()
    import multiprocessing as mp
    seeds = list(range(a.seed0, a.seed0 + a.seeds))
    cells = [("Integrator v1 (as shipped)", "v1", None, None), ("comparator, Bullet's defaults", "si", None, None),
             ("comparator, friction along the plane-space vector only", "si", dict(friction_dir=0), None),
             ("comparator, no friction warm start", "si", dict(friction_warmstart=0), None),
             ("Integrator v1, no link contacts", "v1", None, dict(link_contacts=0)), ("comparator, no link contacts", "si", dict(link_contacts=0), None)]
    lines = ["# tools/a2_long_cycles.py: configs[2] (raptor + narrow_gaps + xavier MACE net), seeds %d..%d x %d frames" % (seeds[0], seeds[-1], a.frames)]
    with mp.get_context("fork").Pool(a.jobs) as pool:
        for tag, integ, so, vo in cells:
            res = pool.map(run_seed, [(integ, sd, a.frames, so, vo) for sd in seeds], chunksize=1)
            cyc = []
            for steps, falls in res:
                cyc += cycles_of(steps, falls)
            lines += describe(tag, cyc)
            print("\n".join(lines[-4:]), flush=True)
    if a.out:
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
