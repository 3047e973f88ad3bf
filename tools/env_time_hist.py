#!/usr/bin/env python3
"""Distribution of the per-env time of ONE frame launch in the bench workload's steady state (DTRL_PROFILE build, lib/libdtrl_prof.so; run via gpurun):
a frame launch has one wavefront per env, all resident at once, so a launch lasts as long as its slowest env. Prints percentiles of the per-env total (ticks of the
100 MHz clock), and what the slow envs have in common (policy forward in this frame, constraint rows, substeps, resets).   tools/env_time_hist.py [envs] [preroll frames] [config]"""
import ctypes as C, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import deepterrainrl_amd as da
import bench
da.LIB_PATH = os.environ.get("DTRL_PROF_LIB") or os.path.join(REPO, "deepterrainrl_amd", "lib", "libdtrl_prof.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 120
cfg = bench.CONFIGS[int(sys.argv[3]) if len(sys.argv) > 3 else 1]
b = da.BatchScenario(cfg["arg_file"], n, data_root=bench.ROOT, extra_args={"terrain_seed": 20260925, "rand_seed": 1})
b.SetPolicy(bench.xavier_weights(b.PolicyNumParams(), cfg["n_char"], cfg["frag"]), *bench.load_scale(cfg))
b.RunFrames(pre)
K = {"total": 13, "rows": 14, "substeps": 15, "nn_evals": 33, "pgs": 8, "detect": 4, "ctrl": 10, "frame_io": 12}
def read():
    out = {}
    buf = (C.c_ulonglong * n)()
    for k, sec in K.items():
        assert b._lib.dtrlx_profile_env(b._h, sec, buf, n) == 0
        out[k] = np.array(list(buf), dtype=np.float64)
    return out
acc = []
pred = []       # (predictor A: the previous frame's constraint-row sum = what EnvStatus::cost carries; predictor B: the row count of the previous frame's LAST substep; this frame's time)
prev_rows = None
for f in range(10):
    r0 = read(); res0 = b.CycleInfo()[1].copy()
    last_R = b.ContactCache()[0].astype(np.float64)       # EnvState::ws_R = rows of the last substep
    b.Update()
    r1 = read(); res1 = b.CycleInfo()[1]
    d = {k: r1[k] - r0[k] for k in K}
    d["reset"] = (res1 != res0).astype(np.float64)
    acc.append(d)
    if prev_rows is not None:
        pred.append((prev_rows, last_R, d["total"]))
    prev_rows = d["rows"] + 8 * d["substeps"]
tot = np.concatenate([d["total"] for d in acc])
print("per-env time of one frame (20 env-steps), %d envs x 10 frames, ticks @100 MHz: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  p99.9 %.0f  max %.0f   (max / mean = %.2f)" % (
    n, tot.mean(), *np.percentile(tot, [50, 90, 99, 99.9]), tot.max(), tot.max() / tot.mean()))
per_frame_max = np.array([d["total"].max() for d in acc]); per_frame_mean = np.array([d["total"].mean() for d in acc])
print("per frame: slowest env / mean env = %s" % " ".join("%.2f" % x for x in per_frame_max / per_frame_mean))
cat = lambda k: np.concatenate([d[k] for d in acc])
ev, rows, sub, rst = cat("nn_evals"), cat("rows"), cat("substeps"), cat("reset")
order = np.argsort(-tot)
for name, sel in (("slowest 1 %", order[:len(order) // 100]), ("all", order)):
    print("%-12s: time %.0f | forward in frame %.2f | rows per substep %.2f | substeps %.1f | reset %.3f | pgs %.0f detect %.0f ctrl %.0f frame-io %.0f" % (
        name, tot[sel].mean(), ev[sel].mean(), rows[sel].sum() / max(sub[sel].sum(), 1), sub[sel].mean(), rst[sel].mean(), cat("pgs")[sel].mean(), cat("detect")[sel].mean(), cat("ctrl")[sel].mean(), cat("frame_io")[sel].mean()))
for lo, hi in ((0, 0.5), (0.5, 0.9), (0.9, 0.99), (0.99, 1.0)):
    sel = order[int(len(order) * (1 - hi)):int(len(order) * (1 - lo))] if hi < 1 else order[:int(len(order) * (1 - lo))]
    print("  time quantile %.2f-%.2f: mean time %.0f, forward %.2f, rows/substep %.2f, reset %.3f" % (lo, hi, tot[sel].mean(), ev[sel].mean(), rows[sel].sum() / max(sub[sel].sum(), 1), rst[sel].mean()))

# VERDICT r5 #5a: which work estimate of frame f - 1 finds the slow envs of frame f? capture = share of the slowest 5 % that the estimate's top 5 % contains
def capture(p, t, q=0.05):
    k = max(1, int(len(t) * q)); top_t = set(np.argsort(-t)[:k]); top_p = set(np.argsort(-p, kind="stable")[:k]); return len(top_t & top_p) / float(k)
def rank_corr(p, t):
    rp = np.argsort(np.argsort(p)); rt = np.argsort(np.argsort(t)); return float(np.corrcoef(rp, rt)[0, 1])
if pred:
    ca = np.mean([capture(a, t) for a, bb, t in pred]); cb = np.mean([capture(bb, t) for a, bb, t in pred])
    ra = np.mean([rank_corr(a, t) for a, bb, t in pred]); rb = np.mean([rank_corr(bb, t) for a, bb, t in pred])
    print("work estimates for the launch order (9 frame pairs): previous frame's row sum (shipped): capture of the slowest 5 %% = %.2f, rank correlation %.2f | rows of the previous frame's LAST substep: capture %.2f, rank correlation %.2f" % (ca, ra, cb, rb))
