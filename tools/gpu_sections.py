#!/usr/bin/env python3
"""Per-section cycle breakdown of the frame kernel (DTRL_PROFILE build, lib/libdtrl_prof.so). Run via gpurun."""
import ctypes as C, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import deepterrainrl_amd as da
import bench
NAMES = ["FK", "Mass", "Bias", "Fact", "Detect", "Rows", "Fsub", "Delassus", "Pgs", "Finish", "Ctrl(incl Action)", "Action", "FrameIO", "Total", "RowsSum", "Substeps", "P1cum", "P2cum", "P3cum", "P4cum", "nR0", "nR1_6", "nR7_12", "nR13_18", "nR19_24", "tR0", "tR1_6", "tR7_12", "tR13_18", "tR19_24", "nnConv", "nnFcTerr", "nnRest", "nnEvals", "cFsm", "cFeedback(+action)", "cPdSetup", "cPdSolve", "cGrav", "cTail"]
da.LIB_PATH = os.environ.get("DTRL_PROF_LIB") or os.path.join(REPO, "deepterrainrl_amd", "lib", "libdtrl_prof.so")   # developer build with section counters
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
CFG = bench.CONFIGS[int(sys.argv[3]) if len(sys.argv) > 3 else 1]
b = da.BatchScenario(CFG["arg_file"], n, data_root=bench.ROOT, extra_args={"terrain_seed": 20260925, "rand_seed": 1, "link_contacts": int(os.environ.get("LINK_CONTACTS", "1"))})
b.SetPolicy(bench.xavier_weights(b.PolicyNumParams(), CFG["n_char"], CFG["frag"]), *bench.load_scale(CFG))
b.RunFrames(int(sys.argv[2]) if len(sys.argv) > 2 else 30)
out = (C.c_ulonglong * 40)()
b._lib.dtrlx_profile_sections(b._h, out, 40)
base = np.array(list(out), dtype=np.float64)
frames = 20
b.KernelTimeMs(); b.RunFrames(frames); ms, nl = b.KernelTimeMs()
b._lib.dtrlx_profile_sections(b._h, out, 40)
v = np.array(list(out), dtype=np.float64) - base
steps = n * frames * 20
print("kernel %.3f ms/frame; per env-step per wave (s_memtime ticks @100MHz -> x24 ~ shader cycles):" % ms)
tot = v[13]
for k, name in enumerate(NAMES):
    if k < 14 or 16 <= k < 20:
        print("  %-18s %10.1f ticks/env-step  %5.1f%%" % (name, v[k] / steps, 100 * v[k] / tot))
print("  avg rows per substep: %.2f" % (v[14] / max(v[15], 1)))
ns = v[20:25]; ts = v[25:30]
for k, nm in enumerate(["R=0", "R 1-6", "R 7-12", "R 13-18", "R 19-24"]):
    print("  substeps with %-8s %5.1f%% of substeps, %5.1f%% of substep time, %8.0f ticks each" % (nm, 100 * ns[k] / max(ns.sum(), 1), 100 * ts[k] / max(ts.sum(), 1), ts[k] / max(ns[k], 1)))
ne = max(v[33], 1)
print("  policy forward: %d evals, per eval: conv %.0f, terr_ip0 %.0f, rest %.0f ticks" % (v[33], v[30] / ne, v[31] / ne, v[32] / ne))
for k in range(34, 40):
    print("  ctrl %-20s %10.1f ticks/env-step" % (NAMES[k], v[k] / steps))
