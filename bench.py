#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched rollout hot path (BASELINE.json metric) on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W [--config 1|2]

N > 1 without a launcher (no WORLD_SIZE in the environment): bench.py starts its own N ranks (python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <same arguments>), one process per GPU over RCCL; under the
driver's own torch.distributed.run line it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual. Rank 0 prints the one JSON line.

Workload (--config 1, default) = BASELINE.json configs[1]: dog + GroundVar2D slopes_mixed, 4096 envs per MI355X, implicit PD + MACE actor/critic
forward (args/dog_slopes_mixed_args.txt; synthetic xavier weights seed 1234 in the dog_mace3_deploy topology -- the trained
*.h5 blobs are not in the reference checkout -- with the shipped dog_mace3_slopes_mixed_model_scale.txt normaliser; fixed-seed
synthetic terrain, seed0 + global env id). --config 2 = configs[2]: raptor + narrow_gaps, 8192 envs per MI355X (raptor_mace3_deploy topology).
One bench "step" = one pass of the hot path over the batch = one outer frame
(cScenarioPoliEval::Update(1/30) on every env) = num_update_steps (20) env-steps per env. 1 env-step = one iteration of the
loop at scenarios/ScenarioSimChar.cpp:162-173 = 1 controller update + 5 physics substeps.
Envs shard across ranks by global env id (weak scaling, no data-path collective: envs are independent).

Timing protocol. The batch starts in lock-step (every env in the same pose and gait phase) and nobody falls for the first ~25 frames, so a
window right behind a short warm-up measures a cheaper workload than the steady state (round 2: 21.3 M in frames 5-25 vs 18.3 M later).
Therefore: (1) an untimed PRE-ROLL steps the batch until the reset rate is stationary (>= 60 frames, independent of --warmup); (2) --warmup W
untimed frames; (3) R REPEATS of a window of EXACTLY --steps K frames, each bracketed by barrier + torch.cuda.synchronize() on both sides with the
max over ranks taken; R is chosen so that the timed GPU work totals >= 1 s. `value` / `ms_per_step` are the MEDIAN window; min / max / repeats are
in the line. Inputs (state, terrain windows, weights) are resident in HBM when a window starts.

`value` is always the pure rollout (weak scaling of the config across N). With --config 1 the line also carries an "exchange" object: a second timed leg
on the BASELINE configs[3] workload (args/opt_args_train_mace.txt: exploration on, rates 0.2 / 0.025 / 0.002) in which every outer frame's experience
tuples are drained device-to-device into a right-sized block, gathered to rank 0 over RCCL on a side stream while the next frame's kernel runs, appended
to a device replay ring there, and the policy is re-broadcast (one packed buffer) every --bcast-every frames -- the two exchange steps of
the north star, measured with the same barrier / synchronize / max-over-ranks bracket. On a multi-rank run the leg runs with one compute unit per XCD
kept free for the collective's kernels (DTRL_RESERVE_CUS=1) and once more without ("exchange_alt"). --exchange-steps 0 skips it.
"""
import os
# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default). The engine's two env-group streams must not share one: with
# RCCL's and the framework's streams in the same process they did (measured: 11.2 M env-steps/s instead of 19.1 M as soon as a process group existed, i.e.
# the two groups' frame kernels serialised). Has to be set before the HIP runtime starts; this is the bench PROCESS's own choice (the package only warns).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import argparse
import json
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ROOT = os.path.join(REPO, "tests", "golden", "refdata")
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s
FP64_VEC_PEAK_TF = 78.6 # the engine computes in fp64 (reference controller/net precision); fp32 vector peak is 157.3
SIMDS = 256 * 4         # MI355X: 256 CUs x 4 SIMD16
CLOCK_GHZ = 2.4         # MI355X_MICROARCH.md peak engine clock
VALU_CYCLES = 4         # one wave64 VALU instruction occupies a SIMD16 for 4 cycles (fp64 FMA is full rate: 78.6 TFLOP/s = 1024 SIMDs x 16 lanes x 2 x 2.4 GHz)
EXCHANGE_ARG_FILE = "args/opt_args_train_mace.txt"
STEPS_PER_FRAME = 20
PREROLL_MIN, PREROLL_MAX, PREROLL_BLOCK = 60, 200, 20
MIN_TIMED_S = 5.0   # (round 5: long enough for the driver's SMI sampler to land inside the timed region; 1 s until round 4)

# B_alg / F_alg: algorithmic bytes / flops per env-step (SURVEY 8d / BASELINE.md 4)
CONFIGS = {
    1: {"arg_file": "args/dog_slopes_mixed_args.txt", "envs": 4096, "n_char": 83, "frag": 29, "b_alg": 1053.0, "f_alg": 0.6e6,
        "scale": "data/policies/dog/models/dog_mace3_slopes_mixed_model_scale.txt", "name": "dog/slopes_mixed", "kernel": "dtrl_frame_kernel_fast<TopoDog>",
        "workload": "BASELINE configs[1]: dog + GroundVar2D slopes_mixed, %d envs per MI355X, ImpPD + MACE actor/critic forward, poli_eval (args/dog_slopes_mixed_args.txt)"},
    2: {"arg_file": "args/raptor_narrow_gaps_args.txt", "envs": 8192, "n_char": 75, "frag": 28, "b_alg": 1059.0, "f_alg": 0.55e6,
        "scale": "data/policies/raptor/models/raptor_mace3_narrow_gaps_model_scale.txt", "name": "raptor/narrow_gaps", "kernel": "dtrl_frame_kernel_fast<TopoRaptor>",
        "workload": "BASELINE configs[2]: raptor + GroundVar2D narrow_gaps, %d envs per MI355X, ImpPD + MACE actor/critic forward, poli_eval (args/raptor_narrow_gaps_args.txt)"},
}


def xavier_weights(num_params_check, n_char=83, frag=29, seed=1234):
    """Same synthetic weights as oracle/model.py xavier_weights (kept here so the timed product leg does not import oracle/)."""
    rng = np.random.RandomState(seed)
    out = []

    def blob(nout, fan_in):
        s = np.sqrt(3.0 / fan_in)
        out.append(rng.uniform(-s, s, size=nout * fan_in).astype(np.float32)); out.append(np.zeros(nout, np.float32))
    blob(16, 8); blob(32, 64); blob(32, 128); blob(64, 5984); blob(256, 64 + n_char); blob(128, 256); blob(3, 128)
    for _ in range(3):
        blob(128, 256); blob(frag, 128)
    w = np.concatenate(out)
    assert w.size == num_params_check
    return w


def load_scale(cfg):
    d = json.load(open(os.path.join(ROOT, cfg["scale"])))
    return [np.asarray(d[k], np.float64) for k in ("InputOffset", "InputScale", "OutputOffset", "OutputScale")]


EXCHANGE_GUARD_S = 180   # wall-clock bound of the exchange leg on a multi-rank run (see main)


def host_parallelism():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU quota where there is one (the GPU boxes report 256 logical CPUs and run
    the job under cpu.max = 16 CPUs: 256 threads there are 16 CPUs' worth of throttled time slices)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                       # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    try:                                                       # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, int(q / p + 0.5)))
    except Exception:
        pass
    return max(1, n)


TRAINED_POLICY = {1: "tests/golden/policies/dog_mace3_slopes_mixed_model.h5", 2: "tests/golden/policies/raptor_mace3_narrow_gaps_model.h5"}   # tools/learn_curve.py, round 5


def trained_policy_leg(da, torch, cfg, n, local_rank, a, path):
    """The same workload under a TRAINED policy (the regime the reference lives in: a character that crosses the terrain instead of stumbling every few cycles). A side
    figure next to the headline, never the headline: the headline keeps the seeded xavier weights every earlier round was measured on."""
    b = da.BatchScenario(cfg["arg_file"], n, data_root=ROOT, device_id=local_rank, extra_args={"terrain_seed": 20260925, "rand_seed": 1, "terrain_gen": a.terrain_gen})
    b.LoadModel(path)                       # Caffe HDF5 weights + <model>_scale.txt (cNeuralNet::LoadModel)
    b.RunFrames(100 + a.warmup)
    st0 = b.EvalStats(); b.KernelTimeMs()
    wins = []
    while sum(wins) < 2.0 and len(wins) < 40:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b.RunFrames(a.steps)
        torch.cuda.synchronize(); wins.append(time.perf_counter() - t0)
    kern_ms, launches = b.KernelTimeMs()
    st1 = b.EvalStats(); b.close()
    dt = float(np.median(wins)); frames = a.steps * len(wins)
    return {"policy": path, "source": "tools/learn_curve.py: 60 000 trainer iterations through this engine (profiles/r05_learning_curve_*.txt)", "env_steps_per_s": n * a.steps * STEPS_PER_FRAME / dt,
            "ms_per_step": dt / a.steps * 1e3, "kernel_avg_ms": kern_ms, "windows": len(wins), "resets_per_frame": (st1["resets"] - st0["resets"]) / float(frames),
            "falls_per_1000_env_steps": 1000.0 * (st1["resets"] - st0["resets"]) / (frames * STEPS_PER_FRAME * float(n)), "policy_forwards_per_frame": (st1["cycles"] - st0["cycles"]) / float(frames)}


def fp32_physics_leg(da, torch, cfg, n, local_rank, a, w, scale):
    """The same workload (same seeds, same xavier weights) on the OPT-IN fp32 build of the kernel source (deepterrainrl_amd/lib/libdtrl_f32.so, -physics_precision= f32):
    `real` = float through physics, controller and policy forward -- half the registers and LDS per env, more wave slots per CU. A side figure, never the headline:
    the headline computes in fp64 (the reference's controller / network precision); this mode holds distribution-level parity only (tests/test_fp32_mode.py)."""
    if not os.path.exists(da.LIB_PATH_F32):
        return {"error": "deepterrainrl_amd/lib/libdtrl_f32.so not built"}
    if a.lib_f32:
        da.LIB_PATH_F32 = os.path.abspath(a.lib_f32)
    b = da.BatchScenario(cfg["arg_file"], n, data_root=ROOT, device_id=local_rank,
                         extra_args={"terrain_seed": 20260925, "rand_seed": 1, "terrain_gen": a.terrain_gen, "physics_precision": "f32"})
    b.SetPolicy(w, *scale)
    b.RunFrames(100 + a.warmup)
    st0 = b.EvalStats(); b.KernelTimeMs()
    wins = []
    while sum(wins) < 2.0 and len(wins) < 40:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b.RunFrames(a.steps)
        torch.cuda.synchronize(); wins.append(time.perf_counter() - t0)
    kern_ms, launches = b.KernelTimeMs()
    st1 = b.EvalStats(); b.close()
    dt = float(np.median(wins)); frames = a.steps * len(wins)
    return {"library": os.path.relpath(da.LIB_PATH_F32, REPO), "dtype": "f32", "note": "opt-in mode (-physics_precision= f32), distribution-level parity only; NOT the headline",
            "env_steps_per_s": n * a.steps * STEPS_PER_FRAME / dt, "ms_per_step": dt / a.steps * 1e3, "kernel_avg_ms": kern_ms, "windows": len(wins),
            "resets_per_frame": (st1["resets"] - st0["resets"]) / float(frames), "falls_per_1000_env_steps": 1000.0 * (st1["resets"] - st0["resets"]) / (frames * STEPS_PER_FRAME * float(n)),
            "policy_forwards_per_frame": (st1["cycles"] - st0["cycles"]) / float(frames)}


def cpu_baseline(cfg, frames=60):
    """The oracle restatement (NOT Bullet -- the reference cannot be built here) timed on the host cores, bounded sample."""
    from oracle import model as om
    m, info = om.build_model(cfg["arg_file"], ROOT)
    desc = om.parse_deploy_prototxt(os.path.join(ROOT, info["args"]["policy_net"]))
    w = om.xavier_weights(desc, 1234)
    io, isc, oo, osc = load_scale(cfg)
    cores = host_parallelism()                  # threads used = CPUs this process can run on at once (cgroup quota respected)
    n_envs = 2048                                # the same bounded sample whatever the thread count
    envs_per_thread = (n_envs + cores - 1) // cores
    n_envs = envs_per_thread * cores
    t0 = time.time()
    rate, resets, cycles = om.batch_run(m, n_envs, cores, frames, terrain_seed0=0, rng_seed=0, policy=(desc, w, io, isc, oo, osc))
    wall = time.time() - t0
    # the single-thread figure beside the all-cores one (one thread, 8 envs, the same frames)
    t1 = time.time()
    rate1, _, _ = om.batch_run(m, 8, 1, frames, terrain_seed0=0, rng_seed=0, policy=(desc, w, io, isc, oo, osc))
    return {"value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs (%d per thread) x %d frames x 20 env-steps of the same workload on the fp64 oracle restatement (not Bullet), %.1f s wall" % (n_envs, envs_per_thread, frames, wall),
            "single_thread": {"value": rate1, "unit": "env-steps/s", "cores": 1, "sample": "%d envs x %d frames x 20 env-steps on one thread, %.1f s wall" % (8, frames, time.time() - t1)},
            "logical_cpus": os.cpu_count() or 1}


class _Dev:
    """Where the batch lives: cuda:<local rank> (the product) or -- --backend emul-tests-only -- the host, with the package's BatchScenario bound to the lane-loop check build
    (tests/emul/libdtrl_emul.so). Keeps the measurement code below free of backend switches."""
    device = None; emul = False; torch = None

    def setup(self, torch, emul, local_rank):
        self.torch, self.emul = torch, emul
        self.device = torch.device("cpu") if emul else torch.device("cuda", local_rank)

    def sync(self):
        if not self.emul:
            self.torch.cuda.synchronize()

    def side_stream(self):
        import contextlib
        if self.emul:
            return contextlib.nullcontext()
        return self.torch.cuda.stream(self.torch.cuda.Stream(device=self.device))

    def package(self, da):
        if not self.emul:
            return da
        lib = os.path.join(REPO, "tests", "emul", "libdtrl_emul.so")

        class EmulScenario(da.BatchScenario):
            def _library(self):
                return da._bind(lib)
        import types
        return types.SimpleNamespace(BatchScenario=EmulScenario)


DEV = _Dev()


def exchange_leg(da, dist, torch, world, rank, local_rank, n, steps, warmup, bcast_every, w, scale, reserve_cus, force_collectives=None):
    """BASELINE configs[3] at this world size: exploration rollouts + the two exchange steps, overlapped with stepping."""
    from deepterrainrl_amd.sharding import ShardedRollout
    dev = DEV.device
    prev = os.environ.get("DTRL_RESERVE_CUS")
    if reserve_cus is not None:
        os.environ["DTRL_RESERVE_CUS"] = str(reserve_cus)      # read by the engine when the batch is created

    def make(n_local, off):
        b = da.BatchScenario(EXCHANGE_ARG_FILE, n_local, data_root=ROOT, device_id=local_rank,
                             extra_args={"terrain_seed": 20260925, "rand_seed": 1, "global_env_offset": off})
        return b
    try:
        sr = ShardedRollout(make, n * world, dist=dist, device=dev, pipelined=True, force_collectives=force_collectives)   # dist None: no collective (the alternative leg of a 1-GPU run)
    finally:
        if reserve_cus is not None:
            if prev is None:
                os.environ.pop("DTRL_RESERVE_CUS", None)
            else:
                os.environ["DTRL_RESERVE_CUS"] = prev
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
        DEV.sync()

    def frame(k):
        nonlocal cursor, tuples
        # Frame k is running (UpdateBegin was called). The engine keeps two tuple rings (dtrl_set_tuple_pipelining): frame k + 1 is launched as soon as
        # frame k has ended, and frame k's tuples are drained, packed and gathered while it runs; the gathered block is consumed one frame later.
        # What stays in the gap between two frame kernels is the frame-boundary host work alone.
        sr.UpdateEndBegin()                              # each env group: frame-boundary host work of frame k, then frame k + 1 at once (its tuples go to the other ring) ...
        if bcast_every > 0 and (k + 1) % bcast_every == 0:
            # the policy hand-over rides beside frame k + 1 as well: weights only (the normalisers have not changed), so every rank parks them in the engine's
            # second weight buffer and frame k + 2's launch switches to it (dtrl_set_policy_device during a frame) -- no barrier left in the loop
            if rank == 0:
                sr.broadcast_policy(pol[0], src=0, normalizers=False, want_host=False)
            else:
                sr.broadcast_policy(src=0, normalizers=False, want_host=False)
        if sr._pending is not None:
            g = sr.gather_tuples_end(dst=0, want_meta=False)   # gather of frame k - 1's tuples: started a whole frame ago
            if rank == 0:
                rows = g[0]; m = int(rows.shape[0])
                if m:                                    # append to the device replay ring: one copy, two when the ring wraps
                    first = min(m, replay_cap - cursor)
                    replay[cursor:cursor + first].copy_(rows[:first])
                    if first < m:
                        replay[:m - first].copy_(rows[first:])
                    cursor = (cursor + m) % replay_cap; tuples += m
        sr.gather_tuples_begin(dst=0)                    # ... while frame k's tuples are drained device-to-device, packed and put on the wire
    # the framework ops of this loop (replay append, count read-backs) go to a stream of their own: the legacy default stream would serialise them with
    # every blocking stream of the process (the engine's CU-masked frame streams are such, DTRL_RESERVE_CUS)
    with DEV.side_stream():
        sr.UpdateBegin()
        for k in range(warmup):
            frame(k)
        b.KernelTimeMs(); sr.exchange_wait_s = 0.0; tuples = 0; sr.carried_rows = 0
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
    res = {"workload": "BASELINE configs[3] shape: dog slopes_mixed, exploration on (0.2 / 0.025 / 0.002), %d envs per GPU x %d GPUs; per outer frame: device tuple drain into a %d-row block -> one gather to rank 0 (RCCL) on a side stream overlapped with the next frame kernel -> device replay ring on rank 0; one packed policy broadcast every %d frames" % (n, world, sr.cap, bcast_every),
           "env_steps_per_s": float(world) * n * steps * 20 / dt, "tuples_per_s": tuples / dt, "tuples": tuples, "steps": steps, "ms_per_step": dt / steps * 1e3,
           "exchange_wait_ms_per_step": sr.exchange_wait_s / steps * 1e3, "tuple_block_rows": sr.cap, "tuple_block_bytes": sr.block_bytes,
           "collective_bytes_per_frame": {"sent_per_rank": sr.block_bytes if sr.coll else 0, "received_by_rank0": sr.block_bytes * (world if sr.coll else 0)},
           "carried_rows": sr.carried_rows, "policy_bytes": int(sr.pol_bytes), "reserve_cus_per_xcd": int(reserve_cus or 0),
           "bcast_every": bcast_every, "dropped_tuples": b.TupleStats()["dropped"] - drop0, "kernel_avg_ms": kern_ms,
           "collective": ("gather to rank 0 (%s)" % ("RCCL" if DEV.device.type == "cuda" else "gloo, tests only") + ("" if world > 1 else " on a one-rank group")) if sr.coll else "none (1 rank: device drain + replay append only)"}
    b.close()
    return res


def self_launch(n, argv):
    """--gpus N > 1 without a launcher: start N ranks of this script on this node (one process per GPU) and pass their output through."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what the host driver supports (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=1, help="BASELINE.json configs[] index of the single-GPU workload: 1 = dog slopes_mixed 4096 envs (the metric's config), 2 = raptor narrow_gaps 8192 envs")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="override the config's batch size")
    ap.add_argument("--repeats", type=int, default=0, help="timed windows of --steps frames (default: as many as make >= 1 s of timed GPU work, 3..25)")
    ap.add_argument("--terrain-gen", choices=["host", "device"], default="host",
                    help="host: the reference's terrain generator streams (bit-exact windows), regenerated by host workers at the frame boundary (default, the parity-tested mode); "
                         "device: counter-based streams, windows generated and slid by the GPU, no host sync per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=60, help="outer frames of the bounded CPU-baseline sample (default: 15-30 s of CPU work on the CPUs the box grants)")
    ap.add_argument("--exchange-steps", type=int, default=-1, help="timed frames of the exchange leg (default: half of --steps, at least 50; 0 = skip)")
    ap.add_argument("--bcast-every", type=int, default=10, help="exchange leg: policy broadcast every K frames")
    ap.add_argument("--no-rccl-leg", action="store_true", help="1-GPU run: do not open a one-rank RCCL group for the exchange leg (the leg then runs without any collective)")
    ap.add_argument("--no-trained-leg", action="store_true", help="skip the side figure under the trained policy (tests/golden/policies)")
    ap.add_argument("--model-args", default="", help="ABLATIONS ONLY: comma-separated overrides of the physics model's creation arguments, e.g. warm_start=0,contact_breaking=0 (the round-4 model); the line then carries config.model_overrides and is not the headline")
    ap.add_argument("--backend", choices=["hip", "emul-tests-only"], default="hip",
                    help="hip = the product (libdtrl.so on the GPUs). emul-tests-only: tests/emul/libdtrl_emul.so (the lane-loop CPU build of the kernel source) over a gloo group -- "
                         "NOT a measurement and not a fallback: it exists so that the N-rank protocol of this script (self-launch, sharding, barriers, max over ranks, exchange "
                         "leg, the one JSON line) runs end to end on a box without GPUs (tests/test_abi.py); the line says so in `backend` and `data`. Needs DTRL_TESTS_ONLY_EMUL=1.")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the side figure on the opt-in fp32 build (lib/libdtrl_f32.so)")
    ap.add_argument("--lib-f32", default="", help="EXPERIMENTS ONLY: another build of the fp32 library for the fp32_physics leg")
    ap.add_argument("--lib", default="", help="EXPERIMENTS ONLY: another build of the HIP library (tools/occupancy_ab.sh: lib/libdtrl_dyn3.so ...); the line then carries config.library and is not the headline")
    ap.add_argument("--preroll-max", type=int, default=PREROLL_MAX, help="upper bound of the untimed pre-roll in frames (tests shorten it)")
    ap.add_argument("--dry-launch", action="store_true", help="every rank prints its placement (rank, local rank, world, global env offset, host threads) as one JSON line and exits: checks the launch path without a GPU")
    a = ap.parse_args()

    emul = a.backend != "hip"
    if emul and os.environ.get("DTRL_TESTS_ONLY_EMUL") != "1":
        sys.exit("--backend emul-tests-only is the CPU check build for tests (set DTRL_TESTS_ONLY_EMUL=1); measurements run on the HIP library only")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cfg = CONFIGS[a.config]
    n = a.envs_per_gpu or cfg["envs"]
    # frame-boundary host workers (terrain regeneration): the ranks of a node share its cores (the engine applies the same rule from LOCAL_WORLD_SIZE)
    host_threads = int(os.environ.get("DTRL_HOST_THREADS", "0")) or max(2, min(16, host_parallelism() // (2 * max(1, local_world))))   # (CPUs the cgroup grants, not logical CPUs: 4 / 8 / 16 threads measured equal at one rank)
    if a.dry_launch:
        rec = json.dumps({"dry_launch": True, "rank": rank, "local_rank": local_rank, "world": world, "gpus_arg": a.gpus, "envs_per_gpu": n,
                          "global_env_offset": rank * n, "device": "cuda:%d" % local_rank, "host_threads": host_threads, "config": a.config})
        sys.stdout.flush()
        os.write(1, (rec + "\n").encode())      # one write per rank: the ranks share the launcher's pipe, and print() hands over text and newline separately
        return
    os.environ.setdefault("DTRL_HOST_THREADS", str(host_threads))

    dist = None
    import torch
    DEV.setup(torch, emul, local_rank)
    force = os.environ.get("DTRL_FORCE_COLLECTIVES") == "1"   # validation hook: a one-rank RCCL group, so that a 1-GPU box runs the collective code path
    if world > 1 or force:
        import torch.distributed as dist
        if not emul:
            torch.cuda.set_device(local_rank)
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        import datetime
        if emul:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=EXCHANGE_GUARD_S + 60))
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=EXCHANGE_GUARD_S + 60))
        world = dist.get_world_size()
    elif torch.cuda.is_available() and not emul:
        torch.cuda.set_device(local_rank)

    # the CPU leg FIRST (round 5): 16 s of host work no longer sit between the timed GPU region and the end of the process, where the driver's SMI samples fell
    cpu_rec = cpu_baseline(cfg, a.cpu_frames) if (rank == 0 and world == 1 and not a.no_cpu_baseline) else None
    import deepterrainrl_amd as da
    if a.lib:
        da.LIB_PATH = os.path.abspath(a.lib)
    da = DEV.package(da)
    model_overrides = dict(kv.split("=", 1) for kv in a.model_args.split(",") if kv)
    b = da.BatchScenario(cfg["arg_file"], n, data_root=ROOT, device_id=local_rank,
                         extra_args=dict({"terrain_seed": 20260925, "rand_seed": 1, "global_env_offset": rank * n, "terrain_gen": a.terrain_gen}, **model_overrides))
    w = xavier_weights(b.PolicyNumParams(), cfg["n_char"], cfg["frag"])
    scale = load_scale(cfg)
    b.SetPolicy(w, *scale)

    def fence():
        if dist is not None:
            dist.barrier()
        DEV.sync()

    def agree(x, op):
        """the same decision on every rank (max / min over ranks)"""
        if dist is None:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device=DEV.device)
        dist.all_reduce(t, op=op)
        return float(t.item())

    # (1) pre-roll to the steady state: blocks of 20 frames until the reset rate of the last block is within 25 % of the block before it (and non-zero)
    preroll = 0; rates = []
    r_prev = b.EvalStats()["resets"]
    while preroll < a.preroll_max:
        b.RunFrames(PREROLL_BLOCK); preroll += PREROLL_BLOCK
        r = b.EvalStats()["resets"]; rates.append((r - r_prev) / float(PREROLL_BLOCK)); r_prev = r
        steady = preroll >= PREROLL_MIN and len(rates) >= 2 and rates[-1] > 0 and abs(rates[-1] - rates[-2]) <= 0.25 * max(rates[-1], rates[-2])
        if agree(0.0 if steady else 1.0, dist.ReduceOp.MAX if dist is not None else None) == 0.0:
            break
    # (2) warm-up, (3) R windows of exactly --steps frames
    b.RunFrames(a.warmup)
    b.KernelTimeMs()   # drop pre-roll / warm-up launches from the kernel-time average
    stats0 = b.EvalStats()
    windows = []
    repeats = a.repeats
    while True:
        fence()
        t0 = time.perf_counter()
        b.RunFrames(a.steps)
        fence()
        dt = time.perf_counter() - t0
        if dist is not None:
            dt = agree(dt, dist.ReduceOp.MAX)
        windows.append(dt)
        if repeats <= 0:
            repeats = int(min(60, max(3, np.ceil(MIN_TIMED_S / max(dt, 1e-6)))))
        if len(windows) >= repeats:
            break
    kern_ms, launches = b.KernelTimeMs()
    stats1 = b.EvalStats()
    dt = float(np.median(windows))
    total_env_steps = float(world) * n * a.steps * STEPS_PER_FRAME
    value = total_env_steps / dt
    B_ALG, F_ALG = cfg["b_alg"], cfg["f_alg"]
    if rank == 0:
        frames_timed = a.steps * len(windows)
        wall = float(sum(windows))
        # the engine splits the batch into env groups (own stream each, no frame barrier between them): a launch covers one group
        env_steps_per_launch = n * frames_timed * STEPS_PER_FRAME / max(launches, 1)
        concurrent = max(1, int(round(launches / float(frames_timed))))
        kern_ms = kern_ms if kern_ms > 0 else float("nan")     # (the lane-loop check build has no kernel clock)
        ach = B_ALG * env_steps_per_launch / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch: PMC counters cannot be read from inside this process; the figure comes from the latest committed
        # rocprofv3 --pmc passes of this same command (tools/gpu_profile.sh -> tools/rocpd_summary.py -> profiles/hbm_traffic_config<N>.json)
        traffic = None; valu_insts = None; traffic_source = None
        tj = os.path.join(REPO, "profiles", "hbm_traffic_config%d.json" % a.config)
        if os.path.exists(tj) and n == cfg["envs"]:
            rec = json.load(open(tj))
            if rec:
                traffic = rec.get("hbm_bytes_per_launch"); valu_insts = rec.get("sq_insts_valu_per_launch")
                traffic_source = "profiles/hbm_traffic_config%d.json <- profiles/%s (rocprofv3 --pmc passes of this command at the same batch size; NOT counters of this run)" % (a.config, rec.get("source"))
        meas_ghz = lanes = None
        cj = os.path.join(REPO, "profiles", "shader_clock_config%d.json" % a.config)
        if os.path.exists(cj):
            crec = json.load(open(cj))
            meas_ghz = crec.get("shader_clock_ghz"); lanes = crec.get("active_lanes_per_valu_instruction")
        resets = stats1["resets"] - stats0["resets"]; cycles = stats1["cycles"] - stats0["cycles"]
        rccl = None
        if dist is not None:
            try:
                rccl = {"ranks": world, "version": "n/a (gloo)" if emul else ".".join(str(x) for x in torch.cuda.nccl.version()), "backend": dist.get_backend()}
            except Exception as exc:
                rccl = {"ranks": world, "version": repr(exc)}
        line = {
            "metric": "env-steps/sec (batched rollout) %s" % cfg["name"], "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not emul else "synthetic; NOT A MEASUREMENT: lane-loop CPU check build over gloo (protocol test)",
            "backend": "hip (libdtrl.so, gfx950)" if not emul else "emul-tests-only (tests/emul/libdtrl_emul.so on the host, gloo)",
            "repeats": len(windows), "value_min": total_env_steps / max(windows), "value_max": total_env_steps / min(windows),
            "window_s": {"median": dt, "min": float(min(windows)), "max": float(max(windows)), "total": wall, "each": [round(float(x), 6) for x in windows]}, "preroll": preroll,
            "config": {"workload": cfg["workload"] % n, "baseline_config_index": a.config,
                       "envs_per_gpu": n, "global_envs": n * world, "env_steps_per_step": n * world * STEPS_PER_FRAME,
                       "substeps_per_env_step": 5, "parallelism": "env-sharded x%d, no data-path collective" % world, "terrain_gen": a.terrain_gen, "link_contacts": 1, "contact_model": "Bullet contact persistence: warm-started ground contact rows (0.85), friction held under an unloaded normal, rows within the breaking threshold (DESIGN 4)" if not model_overrides else "ABLATION", "model_overrides": model_overrides,
                       "host_threads_per_rank": host_threads, "library": a.lib or "deepterrainrl_amd/lib/libdtrl.so", "lds_pad_bytes": int(os.environ.get("DTRL_LDS_PAD", "0"))},
            "rccl": rccl,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": float(traffic) if traffic else None, "traffic_source": traffic_source,
                         "kernel": cfg["kernel"], "kernel_avg_ms": kern_ms, "kernel_launches": launches,
                         "algorithmic_bytes_per_env_step": B_ALG, "env_steps_per_launch": env_steps_per_launch, "concurrent_launches": concurrent,
                         "aggregate_achieved": B_ALG * n * frames_timed * STEPS_PER_FRAME / wall / 1e9,
                         "note": "per-launch figure, HIP events on the engine's own streams over all timed windows (launches of the env groups overlap; aggregate_achieved = all bytes / wall time); the fused path is latency/VALU/LDS-bound, not HBM-bound (SURVEY 8d); companion figure below",
                         "valu": {"achieved": F_ALG * env_steps_per_launch / (kern_ms * 1e-3) / 1e12, "peak": FP64_VEC_PEAK_TF, "unit": "TFLOP/s (fp64 vector)",
                                  "frac": F_ALG * env_steps_per_launch / (kern_ms * 1e-3) / 1e12 / FP64_VEC_PEAK_TF, "algorithmic_flops_per_env_step": F_ALG,
                                  "note": "frac is on the survey's reference-shaped F_alg (comparable across implementations), not on executed instructions; 'executed' below is the utilisation figure",
                                  "executed": None if not valu_insts else {
                                      "sq_insts_valu_per_launch": valu_insts, "per_env_step_per_wave": valu_insts / env_steps_per_launch,
                                      "issue_utilisation_per_launch": valu_insts * VALU_CYCLES / (kern_ms * 1e-3 * CLOCK_GHZ * 1e9 * SIMDS),
                                      "issue_utilisation_wall": valu_insts * VALU_CYCLES * launches / (wall * CLOCK_GHZ * 1e9 * SIMDS),
                                      "assumes": "%d cycles per wave64 VALU instruction on a SIMD16, %d SIMDs, %.1f GHz; counter from the committed PMC pass" % (VALU_CYCLES, SIMDS, CLOCK_GHZ),
                                      # the clock the part actually ran at in the committed counter pass (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration) and the two figures on it
                                      "measured_clock_ghz": meas_ghz,
                                      "issue_utilisation_per_launch_at_measured_clock": None if not meas_ghz else valu_insts * VALU_CYCLES / (kern_ms * 1e-3 * meas_ghz * 1e9 * SIMDS),
                                      "issue_utilisation_wall_at_measured_clock": None if not meas_ghz else valu_insts * VALU_CYCLES * launches / (wall * meas_ghz * 1e9 * SIMDS),
                                      "active_lanes_per_valu_instruction": lanes}}},
            "substeps_per_sec": value * 5, "stats": stats1,
            "timed_window": {"frames": frames_timed, "resets": resets, "cycles": cycles, "seconds": wall,
                             "resets_per_frame": resets / float(frames_timed), "preroll_resets_per_frame": rates,
                             "note": "rank 0's envs over all %d windows: %d falls (terrain regeneration + reset launch each) and %d policy forwards happened inside the timed region; the pre-roll ran until the reset rate was stationary" % (len(windows), resets, cycles)},
        }
    b.close()
    if rank == 0 and world == 1 and not model_overrides and not a.no_trained_leg and not emul:
        pth = os.path.join(REPO, TRAINED_POLICY[a.config])
        if os.path.exists(pth) and n == cfg["envs"]:
            try:
                line["trained_policy"] = trained_policy_leg(da, torch, cfg, n, local_rank, a, pth)
            except Exception as exc:
                line["trained_policy"] = {"error": repr(exc)}
    if rank == 0 and world == 1 and not model_overrides and not a.no_fp32_leg and not emul and n == cfg["envs"]:
        try:
            line["fp32_physics"] = fp32_physics_leg(da, torch, cfg, n, local_rank, a, w, scale)
        except Exception as exc:
            line["fp32_physics"] = {"error": repr(exc)}
    ex_steps = a.exchange_steps if a.exchange_steps >= 0 else max(a.steps // 2, 50)
    legs = []
    if ex_steps > 0 and a.config == 1:
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
        # multi-rank: the collective's kernels need wavefront slots while a frame kernel holds every CU for milliseconds -> one CU per XCD kept free; and the plain setting beside it
        # a plain 1-GPU run (the driver's BENCH line): the headline above ran WITHOUT a process group; the exchange leg now gets a one-rank RCCL group of its
        # own, so that the record exercises the collective code path (gather + broadcast through RCCL) -- the no-group figure stays beside it as exchange_alt
        leg_dist, leg_force = dist, None
        plan = [1, 0] if (dist is not None) else [None]
        if dist is None and world == 1 and not a.no_rccl_leg and not emul:
            try:
                import datetime
                import socket
                import torch.distributed as tdist
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
                tdist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", local_rank),
                                         timeout=datetime.timedelta(seconds=EXCHANGE_GUARD_S + 60))
                leg_dist, leg_force, plan = tdist, True, [1, None]
                line["rccl"] = {"ranks": 1, "version": ".".join(str(x) for x in torch.cuda.nccl.version()), "backend": tdist.get_backend(), "scope": "exchange leg only (the headline ran without a process group)"}
            except Exception as exc:
                line["rccl"] = {"ranks": 1, "error": "one-rank group for the exchange leg: %r" % (exc,)}
        for reserve in plan:
            try:
                one_rank_alt = leg_force and reserve is None            # second leg of the plain 1-GPU run: no group, no reservation (round 3's figure)
                legs.append(exchange_leg(da, None if one_rank_alt else leg_dist, torch, world, rank, local_rank, n, ex_steps, max(a.warmup // 2, 30), a.bcast_every, w, scale, reserve,
                                         force_collectives=None if one_rank_alt else leg_force))   # >= 30 untimed frames: the first tuples complete after two gait cycles (~25 frames)
            except Exception as exc:   # the headline measurement above stands on its own: report the failure instead of losing the line
                legs.append({"error": repr(exc), "reserve_cus_per_xcd": reserve})
        if guard is not None:
            guard.cancel()
    if rank == 0:
        line["exchange"] = legs[0] if legs else None
        if len(legs) > 1:
            line["exchange_alt"] = legs[1]
        if cpu_rec is not None:
            line["cpu_baseline"] = cpu_rec
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)   # RCCL's version banner sits in the C stdio buffer: out with it first, the JSON line is the last line
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    else:
        try:
            import torch.distributed as tdist
            if tdist.is_initialized():
                tdist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
