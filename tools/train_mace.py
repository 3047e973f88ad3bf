#!/usr/bin/env python3
"""Train a MACE policy end to end on the GPU (rollouts + trainer). Example (via gpurun):
   python tools/train_mace.py --arg-file args/opt_args_train_mace.txt --envs 4096 --frames 300"""
import argparse, os, sys
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts (deepterrainrl_amd.configure_hw_queues)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepterrainrl_amd import train_loop
ap = argparse.ArgumentParser()
ap.add_argument("--arg-file", default="args/opt_args_train_mace.txt")
ap.add_argument("--data-root", default=os.path.join(REPO, "tests", "golden", "refdata"))
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--iters", type=int, default=None)
ap.add_argument("--frames-per-drain", type=int, default=1)
ap.add_argument("--overlap", action="store_true", help="train on frame f while frame f+1 rolls out (only pays off when the trainer has its own GPU: on one GPU the frame kernel fills every CU and the trainer's small kernels queue behind it -- measured 4.0 vs 4.7 M env-steps/s)")
ap.add_argument("--trainer", choices=["torch", "hip"], default="hip", help="hip: the MI355X-native trainer step (hip_trainer.py); torch: the PyTorch peer (trainer.py, HIP-graph replay)")
ap.add_argument("--out", default=None, help="write weights (.npy) and <out>_scale.txt")
a = ap.parse_args()
st = train_loop.train(a.arg_file, a.data_root, a.envs, max_iters=a.iters, max_frames=a.frames, log_every=50, overlap=a.overlap, frames_per_drain=a.frames_per_drain,
                      out_scale_file=(a.out + "_scale.txt") if a.out else None, trainer=a.trainer)
if a.out:
    np.save(a.out + ".npy", st["weights"])
print("[trainer=%s] " % a.trainer, end="")
print("frames %d  trainer iters %d  tuples %d  %.1f s  ->  %.2f M env-steps/s while training, %.1f trainer iters/s" % (
    st["frames"], st["iters"], st["tuples"], st["seconds"], st["env_steps_per_s"] / 1e6, st["trainer_iters_per_s"]))
