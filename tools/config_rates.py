#!/usr/bin/env python3
"""Throughput of the BASELINE configurations other than the bench line (SURVEY 8d configs 3 and 5 on one GPU). Run via gpurun."""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import deepterrainrl_amd as da
from deepterrainrl_amd import trainer as tr
ROOT = os.path.join(REPO, "tests", "golden", "refdata")
CASES = [("args/dog_slopes_mixed_args.txt", 4096, "data/policies/dog/nets/dog_mace3_deploy.prototxt", "data/policies/dog/models/dog_mace3_slopes_mixed_model_scale.txt"),
         ("args/raptor_narrow_gaps_args.txt", 8192, "data/policies/raptor/nets/raptor_mace3_deploy.prototxt", "data/policies/raptor/models/raptor_mace3_narrow_gaps_model_scale.txt"),
         ("args/goat_cliffs_args.txt", 8192, None, None)]
PRECISIONS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f64"]     # tools/config_rates.py f64,f32: the opt-in fp32 build beside the product
for arg, n, net, scale, prec in [c + (p,) for c in CASES for p in PRECISIONS]:
    b = da.BatchScenario(arg, n, data_root=ROOT, extra_args={"terrain_seed": 7, "rand_seed": 1, "physics_precision": prec})
    if b.PolicyNumParams():
        if net is None:
            args = {k: v for k, v in (l.strip().lstrip("-").split("= ") for l in open(os.path.join(ROOT, arg)) if "= " in l)}
            net = args["policy_net"]
        w = tr.MaceNet(tr.parse_net(os.path.join(ROOT, net))).init_fillers(1234).get_flat()
        if scale:
            d = json.load(open(os.path.join(ROOT, scale)))
            b.SetPolicy(w, *[np.asarray(d[k], np.float64) for k in ("InputOffset", "InputScale", "OutputOffset", "OutputScale")])
        else:
            b.SetPolicy(w, None, None, *b.BuildNNOutputOffsetScale())
    b.RunFrames(20); b.KernelTimeMs()
    t = time.time(); b.RunFrames(40); dt = time.time() - t
    ms, nl = b.KernelTimeMs()
    print("%-36s %s envs=%5d: %.2f M env-steps/s wall, kernel %.2f ms/frame; %s" % (arg, prec, n, n * 40 * 20 / dt / 1e6, ms, b.EvalStats()), flush=True)
    b.close()
