#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched rollout hot path (BASELINE.json metric) on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

Workload = BASELINE.json configs[1]: dog + GroundVar2D slopes_mixed, 4096 envs per MI355X, implicit PD + MACE actor/critic
forward (args/dog_slopes_mixed_args.txt; synthetic xavier weights seed 1234 in the dog_mace3_deploy topology -- the trained
*.h5 blobs are not in the reference checkout -- with the shipped dog_mace3_slopes_mixed_model_scale.txt normaliser; fixed-seed
synthetic terrain, seed0 + global env id). One bench "step" = one pass of the hot path over the batch = one outer frame
(cScenarioPoliEval::Update(1/30) on every env) = num_update_steps (20) env-steps per env. 1 env-step = one iteration of the
loop at scenarios/ScenarioSimChar.cpp:162-173 = 1 controller update + 5 physics substeps.
Envs shard across ranks by global env id (weak scaling, no data-path collective: envs are independent); the timed region is
bracketed by barrier + torch.cuda.synchronize() on both sides and the max over ranks is taken.
Inputs (state, terrain windows, weights) are resident in HBM when the timed region starts.

`value` is always the pure rollout (weak scaling of BASELINE configs[1] across N). The line also carries an "exchange" object: a second, shorter
timed leg on the BASELINE configs[3] workload (args/opt_args_train_mace.txt: exploration on, rates 0.2 / 0.025 / 0.002) in which every outer
frame's experience tuples are drained device-to-device, all-gathered over RCCL on a side stream while the next frame's kernel runs, appended
to a device replay ring on rank 0, and the policy is re-broadcast (one packed buffer) every --bcast-every frames -- the two exchange steps of
the north star, measured with the same barrier / synchronize / max-over-ranks bracket. --exchange-steps 0 skips it.
"""
import os
# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default). The engine's two env-group streams must not share one: with
# RCCL's and the framework's streams in the same process they did (measured: 11.2 M env-steps/s instead of 19.1 M as soon as a process group existed, i.e.
# the two groups' frame kernels serialised). Has to be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ROOT = os.path.join(REPO, "tests", "golden", "refdata")
ARG_FILE = "args/dog_slopes_mixed_args.txt"
ENVS_PER_GPU = 4096
B_ALG = 1053.0          # algorithmic bytes per env-step, dog (SURVEY 8d / BASELINE.md 4)
F_ALG = 0.6e6           # algorithmic flops per env-step, dog
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s
FP64_VEC_PEAK_TF = 78.6 # the engine computes in fp64 (reference controller/net precision); fp32 vector peak is 157.3
SIMDS = 256 * 4         # MI355X: 256 CUs x 4 SIMD16
CLOCK_GHZ = 2.4         # MI355X_MICROARCH.md peak engine clock
VALU_CYCLES = 4         # one wave64 VALU instruction occupies a SIMD16 for 4 cycles (fp64 FMA is full rate: 78.6 TFLOP/s = 1024 SIMDs x 16 lanes x 2 x 2.4 GHz)
EXCHANGE_ARG_FILE = "args/opt_args_train_mace.txt"


def xavier_weights(num_params_check, seed=1234):
    """Same synthetic weights as oracle/model.py xavier_weights (kept here so the timed product leg does not import oracle/)."""
    rng = np.random.RandomState(seed)
    out = []

    def blob(nout, fan_in):
        s = np.sqrt(3.0 / fan_in)
        out.append(rng.uniform(-s, s, size=nout * fan_in).astype(np.float32)); out.append(np.zeros(nout, np.float32))
    blob(16, 8); blob(32, 64); blob(32, 128); blob(64, 5984); blob(256, 147); blob(128, 256); blob(3, 128)
    for _ in range(3):
        blob(128, 256); blob(29, 128)
    w = np.concatenate(out)
    assert w.size == num_params_check
    return w


def load_scale():
    d = json.load(open(os.path.join(ROOT, "data/policies/dog/models/dog_mace3_slopes_mixed_model_scale.txt")))
    return [np.asarray(d[k], np.float64) for k in ("InputOffset", "InputScale", "OutputOffset", "OutputScale")]


EXCHANGE_GUARD_S = 180   # wall-clock bound of the exchange leg on a multi-rank run (see main)


def cpu_baseline(frames=60):
    """The oracle restatement (NOT Bullet -- the reference cannot be built here) timed on the host cores, bounded sample."""
    from oracle import model as om
    m, info = om.build_model(ARG_FILE, ROOT)
    desc = om.parse_deploy_prototxt(os.path.join(ROOT, info["args"]["policy_net"]))
    w = om.xavier_weights(desc, 1234)
    io, isc, oo, osc = load_scale()
    cores = os.cpu_count() or 1
    envs_per_thread = 8
    n_envs = cores * envs_per_thread
    t0 = time.time()
    rate, resets, cycles = om.batch_run(m, n_envs, cores, frames, terrain_seed0=0, rng_seed=0, policy=(desc, w, io, isc, oo, osc))
    return {"value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs (%d per thread) x %d frames x 20 env-steps of the same workload on the fp64 oracle restatement (not Bullet), %.1f s wall" % (n_envs, envs_per_thread, frames, time.time() - t0)}


def exchange_leg(da, dist, torch, world, rank, local_rank, n, steps, warmup, bcast_every, w, scale):
    """BASELINE configs[3] at this world size: exploration rollouts + the two exchange steps, overlapped with stepping."""
    from deepterrainrl_amd.sharding import ShardedRollout
    dev = torch.device("cuda", local_rank)

    def make(n_local, off):
        b = da.BatchScenario(EXCHANGE_ARG_FILE, n_local, data_root=ROOT, device_id=local_rank,
                             extra_args={"terrain_seed": 20260925, "rand_seed": 1, "global_env_offset": off})
        return b
    sr = ShardedRollout(make, n * world, dist=dist, device=dev, pipelined=True)   # dist is None on a plain 1-GPU run (no collective), a process group otherwise
    b = sr.batch
    if rank == 0:
        sr.broadcast_policy(w, *scale, src=0)
    else:
        sr.broadcast_policy(src=0)
    b.SetExplore(True, 0.2, 0.025, 0.002)        # args/opt_args_train_mace.txt:25-27
    W = b.W
    replay_cap = 1 << 18
    replay = torch.zeros((replay_cap, W), dtype=torch.float32, device=dev) if rank == 0 else None   # device replay ring on the trainer rank
    cursor = 0; tuples = 0
    pol = [v.clone() for v in sr._pol_views()]

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def frame(k):
        nonlocal cursor, tuples
        # Frame k is running (UpdateBegin was called). The engine keeps two tuple rings (dtrl_set_tuple_pipelining): frame k + 1 is launched as soon as
        # frame k has ended, and frame k's tuples are drained, packed and gathered while it runs; the gathered block is consumed one frame later.
        # What stays in the gap between two frame kernels is the frame-boundary host work and, every bcast_every frames, the policy hand-over.
        if bcast_every > 0 and (k + 1) % bcast_every == 0:
            sr.UpdateEnd()                               # frame k finished everywhere: the one place with a barrier, the policy hand-over
            if rank == 0:
                sr.broadcast_policy(*pol, src=0)
            else:
                sr.broadcast_policy(src=0)
            sr.UpdateBegin()
        else:
            sr.UpdateEndBegin()                          # each env group: frame-boundary host work of frame k, then frame k + 1 at once (its tuples go to the other ring) ...
        if sr._pending is not None:
            g = sr.gather_tuples_end(dst=0, want_meta=False)   # all-gather of frame k - 1's tuples: started a whole frame ago
            if rank == 0:
                rows = g[0]; m = int(rows.shape[0])
                if m:                                    # append to the device replay ring: one copy, two when the ring wraps
                    first = min(m, replay_cap - cursor)
                    replay[cursor:cursor + first].copy_(rows[:first])
                    if first < m:
                        replay[:m - first].copy_(rows[first:])
                    cursor = (cursor + m) % replay_cap; tuples += m
        sr.gather_tuples_begin()                         # ... while frame k's tuples are drained device-to-device, packed and put on the wire
    # the framework ops of this loop (replay append, count read-backs) go to a stream of their own: the legacy default stream would serialise them with
    # every blocking stream of the process (the engine's CU-masked frame streams are such, DTRL_RESERVE_CUS)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        sr.UpdateBegin()
        for k in range(warmup):
            frame(k)
        b.KernelTimeMs(); sr.exchange_wait_s = 0.0; tuples = 0
        drop0 = b.TupleStats()["dropped"]
        fence()
        t0 = time.perf_counter()
        for k in range(steps):
            frame(k)
        fence()
        dt = time.perf_counter() - t0
    sr.UpdateEnd()
    if sr._pending is not None:
        sr.gather_tuples_end(dst=0)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms, launches = b.KernelTimeMs()
    return {"workload": "BASELINE configs[3] shape: dog slopes_mixed, exploration on (0.2 / 0.025 / 0.002), %d envs per GPU x %d GPUs; per outer frame: device tuple drain -> one all-gather (RCCL) on a side stream overlapped with the next frame kernel -> device replay ring on rank 0; one packed policy broadcast every %d frames" % (n, world, bcast_every),
            "env_steps_per_s": float(world) * n * steps * 20 / dt, "tuples_per_s": tuples / dt, "tuples": tuples, "steps": steps, "ms_per_step": dt / steps * 1e3,
            "exchange_wait_ms_per_step": sr.exchange_wait_s / steps * 1e3, "tuple_block_bytes": int(sr.block.numel() * 4), "policy_bytes": int(sr.pol_bytes),
            "bcast_every": bcast_every, "dropped_tuples": b.TupleStats()["dropped"] - drop0, "kernel_avg_ms": kern_ms, "collective": ("all_gather (RCCL)" + ("" if world > 1 else " on a one-rank group (DTRL_FORCE_COLLECTIVES)")) if sr.coll else "none (1 rank: device drain + replay append only)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--terrain-gen", choices=["host", "device"], default="host",
                    help="host: the reference's terrain generator streams (bit-exact windows), regenerated by host workers at the frame boundary (default, the parity-tested mode); "
                         "device: counter-based streams, windows generated and slid by the GPU, no host sync per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=60, help="outer frames of the bounded CPU-baseline sample (default: 20-30 s of CPU work on the box's 256 threads)")
    ap.add_argument("--exchange-steps", type=int, default=-1, help="timed frames of the exchange leg (default: half of --steps; 0 = skip)")
    ap.add_argument("--bcast-every", type=int, default=10, help="exchange leg: policy broadcast every K frames")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    force = os.environ.get("DTRL_FORCE_COLLECTIVES") == "1"   # validation hook: a one-rank RCCL group, so that a 1-GPU box runs the collective code path
    if world > 1 or force:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        import datetime
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=EXCHANGE_GUARD_S + 60))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    import deepterrainrl_amd as da
    n = a.envs_per_gpu
    b = da.BatchScenario(ARG_FILE, n, data_root=ROOT, device_id=local_rank,
                         extra_args={"terrain_seed": 20260925, "rand_seed": 1, "global_env_offset": rank * n, "terrain_gen": a.terrain_gen})
    w = xavier_weights(b.PolicyNumParams())
    b.SetPolicy(w, *load_scale())

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    b.RunFrames(a.warmup)
    b.KernelTimeMs()   # drop warm-up launches from the kernel-time average
    stats0 = b.EvalStats()
    fence()
    t0 = time.perf_counter()
    b.RunFrames(a.steps)
    fence()
    dt = time.perf_counter() - t0
    kern_ms, launches = b.KernelTimeMs()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    steps_per_frame = 20
    total_env_steps = float(world) * n * a.steps * steps_per_frame
    value = total_env_steps / dt
    if rank == 0:
        # the engine splits the batch into env groups (own stream each, no frame barrier between them): a launch covers one group
        env_steps_per_launch = n * a.steps * steps_per_frame / max(launches, 1)
        concurrent = max(1, int(round(launches / float(a.steps))))
        ach = B_ALG * env_steps_per_launch / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch: PMC counters cannot be read from inside this process; the figure comes from the latest committed
        # rocprofv3 --pmc passes of this same command (tools/gpu_profile.sh -> tools/rocpd_summary.py -> profiles/hbm_traffic.json)
        traffic = None; valu_insts = None; traffic_source = None
        tj = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(tj) and n == ENVS_PER_GPU:
            rec = json.load(open(tj))
            traffic = rec.get("hbm_bytes_per_launch"); valu_insts = rec.get("sq_insts_valu_per_launch")
            traffic_source = "profiles/hbm_traffic.json <- profiles/%s (rocprofv3 --pmc passes of this command at the same batch size; NOT counters of this run)" % rec.get("source")
        stats1 = b.EvalStats()
        line = {
            "metric": "env-steps/sec (batched rollout) dog/slopes_mixed", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: dog + GroundVar2D slopes_mixed, %d envs per MI355X, ImpPD + MACE actor/critic forward, poli_eval (args/dog_slopes_mixed_args.txt)" % n,
                       "envs_per_gpu": n, "global_envs": n * world, "env_steps_per_step": n * world * steps_per_frame,
                       "substeps_per_env_step": 5, "parallelism": "env-sharded x%d, no data-path collective" % world, "terrain_gen": a.terrain_gen, "link_contacts": 1},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": float(traffic) if traffic else None, "traffic_source": traffic_source,
                         "kernel": "dtrl_frame_kernel", "kernel_avg_ms": kern_ms, "kernel_launches": launches,
                         "algorithmic_bytes_per_env_step": B_ALG, "env_steps_per_launch": env_steps_per_launch, "concurrent_launches": concurrent,
                         "aggregate_achieved": B_ALG * n * a.steps * steps_per_frame / dt / 1e9,
                         "note": "per-launch figure (launches of the env groups overlap; aggregate_achieved = all bytes / wall time); the fused path is latency/VALU/LDS-bound, not HBM-bound (SURVEY 8d); companion figure below",
                         "valu": {"achieved": F_ALG * env_steps_per_launch / (kern_ms * 1e-3) / 1e12, "peak": FP64_VEC_PEAK_TF, "unit": "TFLOP/s (fp64 vector)",
                                  "frac": F_ALG * env_steps_per_launch / (kern_ms * 1e-3) / 1e12 / FP64_VEC_PEAK_TF, "algorithmic_flops_per_env_step": F_ALG,
                                  "note": "frac is on the survey's reference-shaped F_alg (comparable across implementations), not on executed instructions; 'executed' below is the utilisation figure",
                                  "executed": None if not valu_insts else {
                                      "sq_insts_valu_per_launch": valu_insts, "per_env_step_per_wave": valu_insts / env_steps_per_launch,
                                      "issue_utilisation_per_launch": valu_insts * VALU_CYCLES / (kern_ms * 1e-3 * CLOCK_GHZ * 1e9 * SIMDS),
                                      "issue_utilisation_wall": valu_insts * VALU_CYCLES * launches / (dt * CLOCK_GHZ * 1e9 * SIMDS),
                                      "assumes": "%d cycles per wave64 VALU instruction on a SIMD16, %d SIMDs, %.1f GHz; counter from the committed PMC pass" % (VALU_CYCLES, SIMDS, CLOCK_GHZ)}}},
            "substeps_per_sec": value * 5, "stats": stats1,
            "timed_window": {"resets": stats1["resets"] - stats0["resets"], "cycles": stats1["cycles"] - stats0["cycles"], "seconds": dt,
                             "note": "characters driven by the synthetic seeded policy fall; falls (terrain regeneration + reset launches) are part of the timed work"},
        }
    ex_steps = a.exchange_steps if a.exchange_steps >= 0 else max(a.steps // 2, 1)
    ex = None
    if ex_steps > 0:
        b.close()
        guard = None
        if rank == 0 and dist is not None:
            # a collective that never completes (a rank died mid-exchange) must not cost the headline measurement above: after EXCHANGE_GUARD_S
            # seconds rank 0 prints the line with the failure recorded and leaves; the other ranks end at the process group's timeout
            import threading

            def give_up():
                line["exchange"] = {"error": "exchange leg did not finish within %d s" % EXCHANGE_GUARD_S}
                print(json.dumps(line), flush=True)
                os._exit(0)
            guard = threading.Timer(EXCHANGE_GUARD_S, give_up); guard.daemon = True; guard.start()
        try:
            ex = exchange_leg(da, dist, torch, world, rank, local_rank, n, ex_steps, max(a.warmup // 2, 30), a.bcast_every, w, load_scale())   # >= 30 untimed frames: the first tuples complete after two gait cycles (~25 frames)
        except Exception as exc:   # the headline measurement above stands on its own: report the failure instead of losing the line
            ex = {"error": repr(exc)}
        if guard is not None:
            guard.cancel()
    if rank == 0:
        line["exchange"] = ex
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.cpu_frames)
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)   # RCCL's version banner sits in the C stdio buffer: out with it first, the JSON line is the last line
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
