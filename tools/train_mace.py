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
ap.add_argument("--overlap", action="store_true", help="train on frame f while frame f+1 rolls out: frame f+1 is relaunched before frame f's tuples are drained (host-memory tuple rings), the weights are parked for the next launch (dog, native trainer, one GPU: 9.1 M sequential, 11.2-11.5 M overlapped; with round 2's PyTorch trainer it did not pay: 4.0 vs 4.7 M)")
ap.add_argument("--trainer", choices=["torch", "hip"], default="hip", help="hip: the MI355X-native trainer step (hip_trainer.py); torch: the PyTorch peer (trainer.py, HIP-graph replay)")
ap.add_argument("--reserve-cus", type=int, default=None, help="compute units per XCD kept out of the frame launches for the trainer's kernels (engine arg -reserve_cus=; default 0: measured 11.3 M env-steps/s without, 9.7 M with 2 per XCD -- the trainer's small GEMMs run 3x slower on 16-32 units than in the frame kernel's gaps on 256)")
ap.add_argument("--init-samples", type=int, default=None, help="override -trainer_num_init_samples= (the arg files collect 50 000 tuples before the first iteration: ~235 frames of 4096 dogs)")
ap.add_argument("--poll", action="store_true", help="with --overlap: relaunch env groups between Train() calls (dtrl_step_poll; measured: no gain on one GPU)")
ap.add_argument("--distributed", action="store_true", help="train_loop.train_distributed: sharded rollout + tuple gather to rank 0 + policy broadcast (RCCL). Start with torch.distributed.run for N ranks; alone it runs a one-rank RCCL group (DTRL_FORCE_COLLECTIVES=1), the config-3/4 loop shape on one GPU")
ap.add_argument("--data-parallel", action="store_true", help="with --distributed: no trainer rank -- every rank trains on its own tuples, gradients all-reduced (train_distributed(mode=\"data_parallel\"))")
ap.add_argument("--out", default=None, help="write weights (.npy) and <out>_scale.txt")
a = ap.parse_args()
reserve = a.reserve_cus if a.reserve_cus is not None else (1 if (a.distributed and not a.data_parallel) else 0)   # (the exchange's collective and read-backs beside the rollout: 11.0 M with one unit per XCD set aside, 9.6 M without)
if a.distributed:
    import torch, torch.distributed as dist
    os.environ.setdefault("DTRL_FORCE_COLLECTIVES", "1")
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29541"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
        os.environ.setdefault(k, v)
    lr = int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(lr)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", lr))
    world = dist.get_world_size()
    ea = dict(({"trainer_num_init_samples": a.init_samples} if a.init_samples is not None else {}), **({"reserve_cus": reserve} if reserve else {})) or None
    st = train_loop.train_distributed(a.arg_file, a.data_root, a.envs * world, dist, max_iters=a.iters, max_frames=a.frames, extra_args=ea, device="cuda:%d" % lr,
                                      trainer_device="cuda:%d" % lr, local_device_id=lr, trainer=a.trainer, overlap=a.overlap, mode="data_parallel" if a.data_parallel else "gather")
    if dist.get_rank() == 0:
        print("[distributed x%d%s, trainer=%s, overlap=%s] frames %d  trainer iters %d  tuples %d  %.1f s  ->  %.2f M env-steps/s while training (all ranks), %.1f trainer iters/s" % (
            world, " data-parallel" if a.data_parallel else "", a.trainer, a.overlap, st["frames"], st["iters"], st["tuples"], st["seconds"], st["env_steps_per_s"] / 1e6, st["iters"] / st["seconds"]))
        if "phases" in st:
            print("   [distributed] rank 0 host wall-clock by phase (ms per frame): " + "  ".join("%s %.2f" % (k, 1e3 * v / max(st["frames"], 1)) for k, v in st["phases"].items()))
    dist.barrier(); dist.destroy_process_group()
    sys.exit(0)
st = train_loop.train(a.arg_file, a.data_root, a.envs, max_iters=a.iters, max_frames=a.frames, log_every=50, overlap=a.overlap, frames_per_drain=a.frames_per_drain,
                      extra_args=dict(({"reserve_cus": reserve} if reserve else {}), **({"trainer_num_init_samples": a.init_samples} if a.init_samples is not None else {})) or None,
                      out_scale_file=(a.out + "_scale.txt") if a.out else None, trainer=a.trainer, poll=a.poll)
if a.out:
    np.save(a.out + ".npy", st["weights"])
print("[trainer=%s reserve_cus=%d side-stream start delay %.0f us] " % (a.trainer, reserve, st["side_stream_delay_us"]), end="")
print("frames %d  trainer iters %d  tuples %d  %.1f s  ->  %.2f M env-steps/s while training, %.1f trainer iters/s" % (
    st["frames"], st["iters"], st["tuples"], st["seconds"], st["env_steps_per_s"] / 1e6, st["trainer_iters_per_s"]))
print("   host wall-clock by phase (ms per frame): " + "  ".join("%s %.2f" % (k, 1e3 * v / max(st["frames"], 1)) for k, v in st["phases"].items() if k != "early_relaunches") + "  | env groups relaunched between Train() calls: %d" % st["phases"].get("early_relaunches", 0))
