"""cScenarioTrain for the batched engine: exploration rollouts on the GPU feed the GPU MACE trainer, weights flow back.

Mirrors scenarios/ScenarioTrain.cpp: ParseArgs (:44-73, the -trainer_* / -init_exp_* / *_anneal_iters keys of
args/opt_args_train_mace.txt), InitTrainer + SetupTrainerOutputOffsetScale (:251-257, 322-338), UpdateExpScene (:369-404: when a
scene's tuple buffer is full -> learner->Train(tuples) -> annealed exploration parameters and curriculum back into the scene).
The reference runs one OS thread per scene, each handing 32 tuples (-tuple_buffer_size=) to the trainer per call; here all scenes
advance together, the drained rows are fed in chunks of that size, and the policy is re-synchronised once per outer frame."""
import os
import re
import time

import numpy as np

from . import BatchScenario
from .trainer import CaclaTrainer, MACETrainer, QNetTrainer, anneal


def parse_arg_file(path):
    """util/ArgParser.cpp:42-108 token format: -key= value (comments start with #)."""
    out = {}
    toks = re.sub(r"#[^\n]*", "", open(path).read()).split()
    i = 0
    while i < len(toks):
        if toks[i].startswith("-") and toks[i].endswith("="):
            key = toks[i][1:-1]; i += 1
            vals = []
            while i < len(toks) and not (toks[i].startswith("-") and toks[i].endswith("=")):
                vals.append(toks[i]); i += 1
            out[key] = " ".join(vals)
        else:
            i += 1
    return out


def _make_trainer(args, data_root, train_net, solver, b, tkw, trainer="torch", trainer_lib=None):
    """cScenarioTrain::BuildTrainer and its subclasses: *_mace controllers train with cMACETrainer, -char_ctrl= dog / raptor (the Q controllers,
    scenarios/ScenarioSimChar.cpp:421-430) with cQNetTrainer, *_cacla with cCaclaTrainer (scenarios/ScenarioTrainCacla.cpp:21-52: -policy_* name the
    ACTOR, -critic_* the critic the trainer steps first)."""
    ctrl = args.get("char_ctrl", "")
    if ctrl.endswith("_cacla"):
        c_solver = os.path.join(data_root, args["critic_solver"])
        m = re.search(r'net:\s*"([^"]+)"', open(c_solver).read())
        c_train = os.path.join(data_root, m.group(1)) if m else os.path.join(data_root, args["critic_net"].replace("_deploy", "_train"))
        if trainer == "hip":
            from .hip_trainer import HipCaclaTrainer
            return HipCaclaTrainer(c_train, c_solver, train_net, solver, b.S, b.A, lib_path=trainer_lib, **tkw)
        return CaclaTrainer(c_train, c_solver, train_net, solver, b.S, b.A, **tkw)
    if ctrl in ("dog", "raptor"):
        if trainer == "hip":
            from .hip_trainer import HipQNetTrainer
            return HipQNetTrainer(train_net, solver, b.S, b.A, lib_path=trainer_lib, **tkw)
        return QNetTrainer(train_net, solver, b.S, b.A, **tkw)
    if trainer == "hip":   # the MI355X-native step (hip_trainer.py): cMACETrainer's iteration on hand-written HIP kernels
        from .hip_trainer import HipMACETrainer
        return HipMACETrainer(train_net, solver, b.S, b.A, lib_path=trainer_lib, **tkw)
    return MACETrainer(train_net, solver, b.S, b.A, **tkw)


def feed_chunks(t, rows, flags, chunk, ph=None, poll=None):
    """cScenarioTrain's schedule: one AddTuples + Train per `chunk` (-tuple_buffer_size=) tuples, in the order given. A native trainer gets the rows
    through its page-locked staging area (written once, stored chunk by chunk from there)."""
    clk = time.perf_counter
    stage = getattr(t, "StageTuples", None)
    base = staged = 0
    for k in range(0, len(rows), chunk):
        c0 = clk()
        if stage is not None:
            if k + min(chunk, len(rows) - k) > base + staged:
                base = k; staged = stage(rows[k:], flags[k:])
            t.AddTuples(rows[k:k + chunk], flags[k:k + chunk], staged=k - base)
        else:
            t.AddTuples(rows[k:k + chunk], flags[k:k + chunk])
        c1 = clk()
        t.Train()
        c2 = clk()
        if poll is not None:
            poll()                      # an env group whose frame ended meanwhile is relaunched now, not when the whole batch of tuples is through
        if ph is not None:
            ph["add_tuples"] += c1 - c0; ph["train"] += c2 - c1
            if poll is not None:
                ph["rollout"] += clk() - c2
    return len(rows)


def train(arg_file, data_root, num_envs=4096, max_iters=None, max_frames=None, device_id=-1, extra_args=None, seed=0, log_every=0, out_scale_file=None,
          trainer_device=None, overlap=False, frames_per_drain=1, scenario_cls=BatchScenario, trainer="torch", trainer_lib=None, poll=False,
          eval_every=None, eval_fn=None, out_model_file=None):
    """extra_args override / extend the arg file (both for the engine and for the -trainer_* keys read here).
    overlap=True trains on frame f's tuples while the GPU already rolls out frame f+1 (dtrl_step_begin / dtrl_step_end): the policy
    each frame runs with is one frame staler, as with the reference's concurrent env threads; overlap=False is the strictly
    sequential, reproducible schedule. frames_per_drain > 1 rolls out several outer frames per drain / policy sync (RunFrames: the env
    groups then run without a frame barrier between them, which is where the rollout engine is fastest). poll=True (with overlap) relaunches an env group
    whose frame ends while the trainer is busy between two Train() calls (dtrl_step_poll) -- measured on one GPU: ~2 early relaunches per frame and no gain
    (dog 9.95-11.2 M with, 11.2-11.4 M without: the trainer's kernels and the frame waves share the same wavefront slots either way), so it is off by default.
    eval_every / eval_fn: eval_fn(iteration, trainer, batch) is called before the first iteration and then whenever the iteration counter has passed another
    eval_every (cScenarioTrain's intermediate output every trainer_int_iter, scenarios/ScenarioTrain.cpp:376-410, with an evaluation in its place: tools/learn_curve.py).
    out_model_file: the trainer's net as a Caffe HDF5 model at the end (cNeuralNetTrainer::OutputModel), next to out_scale_file."""
    if overlap:
        extra_args = dict({"tuple_ring": "host"}, **(extra_args or {}))    # drains beside a running frame must not queue copies behind it (include/dtrl.h: dtrl_drain_tuples)
    args = parse_arg_file(os.path.join(data_root, arg_file))
    args.update({k: str(v) for k, v in (extra_args or {}).items()})
    geti = lambda k, d: int(args.get(k, d)); getf = lambda k, d: float(args.get(k, d))
    b = scenario_cls(arg_file, num_envs, data_root=data_root, device_id=device_id, extra_args=extra_args)
    solver = os.path.join(data_root, args["policy_solver"])
    train_net = os.path.join(data_root, re.search(r'net:\s*"([^"]+)"', open(solver).read()).group(1)) if re.search(r'net:\s*"', open(solver).read()) \
        else os.path.join(data_root, args["policy_net"].replace("_deploy", "_train"))
    tkw = dict(mem_size=geti("trainer_replay_mem_size", 500000), num_init_samples=geti("trainer_num_init_samples", 200),
               steps_per_iter=geti("trainer_num_steps_per_iters", 1), freeze_target_iters=geti("trainer_freeze_target_iters", 0),
               init_input_offset_scale=args.get("trainer_init_input_offset_scale", "false").lower() == "true", seed=seed, device=trainer_device)
    t = _make_trainer(args, data_root, train_net, solver, b, tkw, trainer, trainer_lib)
    t.SetOutputOffsetScale(*b.BuildNNOutputOffsetScale())
    side = b.SideStream(0) if hasattr(b, "SideStream") else (None, -1.0)
    if side[0] is not None and hasattr(t, "UseStream"):
        t.UseStream(side[0])     # (-reserve_cus= k in extra_args: the trainer's kernels run beside the rollout on the reserved compute units)
    exp = dict(rate=getf("exp_rate", 0.2), temp=getf("exp_temp", 0.025), base=getf("exp_base_rate", 0.002))
    init = dict(rate=getf("init_exp_rate", 1.0), temp=getf("init_exp_temp", 20.0), base=getf("init_exp_base_rate", 1.0))
    n_anneal, n_base_anneal = geti("trainer_num_anneal_iters", 1), geti("exp_base_anneal_iters", 1)
    n_curr = geti("trainer_curriculum_iters", 0)
    chunk = max(1, geti("tuple_buffer_size", 32))
    max_iters = max_iters if max_iters is not None else geti("trainer_max_iter", 10 ** 9)

    last_norm = [None]

    def sync(it):
        if hasattr(t, "WeightsDevicePtr"):
            # native trainer: the weights go from the trainer's device buffer into the engine's layout by a gather kernel (dtrl_set_policy_device); the
            # normalisers only travel (from the host) when they changed
            nt = getattr(t, "policy_nt", t.nt)      # (CACLA: the actor's)
            norm = t.GetOffsetScale()
            if last_norm[0] is None or any(not np.array_equal(a, c) for a, c in zip(norm, last_norm[0])):
                nt.sync()
                b.SetPolicy(t.GetWeights(), *norm); last_norm[0] = norm
            else:
                owner = getattr(t, "actor", t) if hasattr(t, "policy_nt") else t
                sp = owner.StreamPtr() if hasattr(owner, "StreamPtr") else None
                if sp is not None and hasattr(b, "SetPolicyDeviceOn"):
                    b.SetPolicyDeviceOn(t.WeightsDevicePtr(), nt.num_params, sp)     # behind the trainer's queued work, one wait for both
                else:
                    nt.sync()
                    b.SetPolicyDevice(t.WeightsDevicePtr(), nt.num_params)
        else:
            b.SetPolicy(t.GetWeights(), *t.GetOffsetScale())
        b.SetExplore(1, anneal(it, n_anneal, init["rate"], exp["rate"]), anneal(it, n_anneal, init["temp"], exp["temp"]), anneal(it, n_base_anneal, init["base"], exp["base"]))
        phase = 1.0 if n_curr < 1 else min(max(it / float(n_curr), 0.0), 1.0)    # CalcCurriculumPhase (gInitCurriculumPhase at iter 0)
        b.SetTerrainParamsLerp(phase if it > 0 or n_curr < 1 else 0.0)

    sync(0)
    frames = tuples = 0
    next_eval = [0 if eval_fn else None]
    if eval_fn:
        eval_fn(0, t, b); next_eval[0] = eval_every
    t0 = time.time()
    stats = {"log": []}
    ph = {"rollout": 0.0, "drain": 0.0, "policy_sync": 0.0, "add_tuples": 0.0, "train": 0.0}   # host wall-clock seconds by phase (stats["phases"])
    clk = time.perf_counter

    def feed(rows, flags, ids, poll=None):
        o = np.argsort(ids, kind="stable")   # the device ring is filled in completion order; env-id order makes the run reproducible and shard-invariant
        return feed_chunks(t, rows[o], flags[o], chunk, ph, poll)

    def log():
        if next_eval[0] is not None and t.GetIter() >= next_eval[0]:
            eval_fn(t.GetIter(), t, b)
            next_eval[0] = (t.GetIter() // eval_every + 1) * eval_every if eval_every else None
        if log_every and frames % log_every == 0:
            stats["log"].append((frames, t.GetIter(), t.GetNumTuples(), t.last_loss, b.EvalStats()))
            print("frame %d iter %d tuples %d critic-loss %s actor-iters %d" % (frames, t.GetIter(), t.GetNumTuples(), t.last_loss, t.actor_iter), flush=True)

    if not overlap:
        while t.GetIter() < max_iters and (max_frames is None or frames < max_frames):
            c0 = clk()
            if frames_per_drain > 1:
                b.RunFrames(frames_per_drain)
            else:
                b.Update(1.0 / 30.0)
            frames += frames_per_drain
            c1 = clk()
            drained = b.DrainTuples()
            ph["rollout"] += c1 - c0; ph["drain"] += clk() - c1
            n = feed(*drained)
            tuples += n
            if n:
                c0 = clk(); sync(t.GetIter()); ph["policy_sync"] += clk() - c0
            log()
    else:
        # frame f+1 is launched the moment frame f's boundary work is done (UpdateEndBegin); frame f's tuples are drained from the idle ring, trained on, and the
        # weights handed over while f+1 runs -- they take effect with frame f+2's launch (dtrl_set_policy_device during a frame): frame f+2 runs on tuples <= f
        want_poll = poll
        b.SetTuplePipelining(True)
        b.UpdateBegin(1.0 / 30.0)
        more = True
        while more:
            frames += 1
            more = t.GetIter() < max_iters and (max_frames is None or frames < max_frames)
            c0 = clk()
            if more:
                b.UpdateEndBegin(1.0 / 30.0)                # frame f's boundary work, then frame f+1 runs on the engine's streams ...
            else:
                b.UpdateEnd()
            c1 = clk()
            drained = b.DrainTuples()                       # ... while frame f's tuples are read from the ring the kernels no longer write,
            c2 = clk()
            # (a group that finishes its frame while the trainer is busy is relaunched between two Train() calls -- only when another UpdateEndBegin is certain to follow)
            sure = more and (max_frames is None or frames + 1 < max_frames) and t.GetIter() + len(drained[0]) // chunk + 2 < max_iters
            def poll_fn():
                ph["early_relaunches"] = ph.get("early_relaunches", 0) + b.UpdatePoll(1.0 / 30.0)
            n = feed(*drained, poll=poll_fn if (want_poll and sure and hasattr(b, "UpdatePoll")) else None)   # the trainer works through them,
            tuples += n
            c3 = clk()
            if n:
                sync(t.GetIter())                           # and the new weights are parked for the next launch
            ph["rollout"] += c1 - c0; ph["drain"] += c2 - c1; ph["policy_sync"] += clk() - c3
            log()
        b.SetTuplePipelining(False)
    dt = time.time() - t0
    if out_scale_file:
        b.WriteOffsetScale(out_scale_file)
    if out_model_file:
        t.OutputModel(out_model_file)
    stats.update(frames=frames, iters=t.GetIter(), tuples=tuples, seconds=dt, env_steps_per_s=frames * 20.0 * num_envs / dt,
                 trainer_iters_per_s=t.GetIter() / dt, weights=t.GetWeights(), offset_scale=t.GetOffsetScale(), phases=ph, side_stream_delay_us=side[1])
    return stats


def train_data_parallel(arg_file, data_root, global_envs, dist, max_iters=None, max_frames=None, extra_args=None, seed=0, trainer_device=None,
                        local_device_id=-1, scenario_cls=BatchScenario, trainer_lib=None, overlap=False):
    """cScenarioTrain over several GPUs WITHOUT a trainer rank (train_distributed(mode="data_parallel")): every rank rolls out its shard of the global env ids,
    keeps ITS OWN tuples in its own replay memory, and steps hip_trainer.HipMACETrainerDP -- two gradient all-reduces per Train() (2.28 MB each) are the only exchange;
    no tuple gather, no weight broadcast: the weights are equal on all ranks because the updates are. Each rank hands its own trainer's weights to its own rollout
    engine (device pointer, no communication). Per outer frame the ranks agree (one all-reduce MAX of an integer) on how many Train() calls the frame gets --
    max over ranks of (new tuples // tuple_buffer_size), the reference's one-Train-per-32-tuples schedule applied to the busiest rank -- so that every rank issues the
    same sequence of collectives; a rank with fewer new tuples still trains (a minibatch is drawn from the whole replay memory, new or not).
    What a rank's trainer sees is 1 / world of the experience: the replay memory, the critic / actor buffers and the minibatch draws are per rank. It is NOT the
    gathered form's tuple stream (train_distributed's default mode reproduces train() bit for bit; this mode trades that for a trainer that scales with the ranks)."""
    import torch
    from .hip_trainer import HipMACETrainerDP
    from .sharding import shard_range
    from . import sharding as _sh
    args = parse_arg_file(os.path.join(data_root, arg_file))
    args.update({k: str(v) for k, v in (extra_args or {}).items()})
    geti = lambda k, d: int(args.get(k, d)); getf = lambda k, d: float(args.get(k, d))
    if not args.get("char_ctrl", "").endswith("_mace"):
        raise ValueError("the data-parallel step is built for the MACE trainers (-char_ctrl= *_mace)")
    rank, world = dist.get_rank(), dist.get_world_size()
    off, n_local = shard_range(global_envs, world, rank)
    ea = dict(extra_args or {}); ea["global_env_offset"] = off
    if overlap:
        ea.setdefault("tuple_ring", "host")
    b = scenario_cls(arg_file, n_local, data_root=data_root, device_id=local_device_id, extra_args=ea)
    solver = os.path.join(data_root, args["policy_solver"])
    m = re.search(r'net:\s*"([^"]+)"', open(solver).read())
    train_net = os.path.join(data_root, m.group(1)) if m else os.path.join(data_root, args["policy_net"].replace("_deploy", "_train"))
    t = HipMACETrainerDP(train_net, solver, b.S, b.A, dist=dist, lib_path=trainer_lib, mem_size=geti("trainer_replay_mem_size", 500000),
                         num_init_samples=max(1, geti("trainer_num_init_samples", 200) // world),      # the file's count is for the whole experience stream
                         steps_per_iter=geti("trainer_num_steps_per_iters", 1), freeze_target_iters=geti("trainer_freeze_target_iters", 0),
                         init_input_offset_scale=args.get("trainer_init_input_offset_scale", "false").lower() == "true", seed=seed + 7919 * rank, device=trainer_device)
    # identical initial weights: rank 0's
    w = torch.from_numpy(np.ascontiguousarray(t.GetWeights(), np.float32))
    wd = w.to(t.device) if t.device.type == "cuda" else w
    _sh.broadcast(dist, wd, 0)
    t.SetWeights(wd.cpu().numpy())
    t.SetOutputOffsetScale(*b.BuildNNOutputOffsetScale())
    exp = dict(rate=getf("exp_rate", 0.2), temp=getf("exp_temp", 0.025), base=getf("exp_base_rate", 0.002))
    init = dict(rate=getf("init_exp_rate", 1.0), temp=getf("init_exp_temp", 20.0), base=getf("init_exp_base_rate", 1.0))
    n_anneal, n_base_anneal, n_curr = geti("trainer_num_anneal_iters", 1), geti("exp_base_anneal_iters", 1), geti("trainer_curriculum_iters", 0)
    chunk = max(1, geti("tuple_buffer_size", 32))
    max_iters = max_iters if max_iters is not None else geti("trainer_max_iter", 10 ** 9)
    last_norm = [None]

    def sync(it):
        norm = t.GetOffsetScale()
        if last_norm[0] is None or any(not np.array_equal(a, c) for a, c in zip(norm, last_norm[0])):
            t.nt.sync()
            b.SetPolicy(t.GetWeights(), *norm); last_norm[0] = norm
        else:
            sp = t.StreamPtr()
            if sp is not None and hasattr(b, "SetPolicyDeviceOn"):
                b.SetPolicyDeviceOn(t.WeightsDevicePtr(), t.nt.num_params, sp)
            else:
                t.nt.sync(); b.SetPolicyDevice(t.WeightsDevicePtr(), t.nt.num_params)
        b.SetExplore(1, anneal(it, n_anneal, init["rate"], exp["rate"]), anneal(it, n_anneal, init["temp"], exp["temp"]), anneal(it, n_base_anneal, init["base"], exp["base"]))
        phase = 1.0 if n_curr < 1 else min(max(it / float(n_curr), 0.0), 1.0)
        b.SetTerrainParamsLerp(phase if it > 0 or n_curr < 1 else 0.0)

    sync(0)
    frames = tuples = 0
    carry = 0          # new tuples not yet accounted for by a Train() call (the reference trains once per full scene buffer)
    t0 = time.time()
    count = torch.zeros(1, dtype=torch.int64, device=t.device if t.device.type == "cuda" else "cpu")
    if overlap:
        b.SetTuplePipelining(True); b.UpdateBegin(1.0 / 30.0)
    more = True
    while more:
        frames += 1
        more = t.GetIter() < max_iters and (max_frames is None or frames < max_frames)
        if overlap:
            b.UpdateEndBegin(1.0 / 30.0) if more else b.UpdateEnd()
        else:
            b.Update(1.0 / 30.0)
        rows, flags, ids = b.DrainTuples()
        o = np.argsort(ids, kind="stable")
        rows, flags = rows[o], flags[o]
        if len(rows):
            k = 0
            while k < len(rows):
                st = t.StageTuples(rows[k:], flags[k:])
                for j in range(0, st, chunk):
                    t.AddTuples(rows[k + j:k + j + chunk], flags[k + j:k + j + chunk], staged=j)
                k += st
        tuples += len(rows); carry += len(rows)
        count[0] = carry // chunk
        _sh.all_reduce(dist, count, op=dist.ReduceOp.MAX)
        n_train = int(count.item())
        carry = max(0, carry - n_train * chunk)
        for _ in range(n_train):
            t.Train()
        if n_train:
            sync(t.GetIter())
    if overlap:
        b.SetTuplePipelining(False)
    dt = time.time() - t0
    return dict(frames=frames, iters=t.GetIter(), actor_iters=t.actor_iter, seconds=dt, env_steps_per_s=frames * 20.0 * global_envs / dt, rank=rank, batch=b, tuples=tuples,
                weights=t.GetWeights(), offset_scale=t.GetOffsetScale(), trainer=t)


def train_distributed(arg_file, data_root, global_envs, dist, max_iters=None, max_frames=None, extra_args=None, seed=0, device=None,
                      trainer_device=None, local_device_id=-1, scenario_cls=BatchScenario, trainer="torch", trainer_lib=None, overlap=False, mode="gather", block_rows=None):
    """cScenarioTrain over several GPUs: one process per GPU (torch.distributed already initialised; backend "nccl" = RCCL on GPUs).
    Every rank rolls out its contiguous range of global env ids; each outer frame the ranks' drained MACE rows are gathered on rank 0
    (the only exchange on the experience side), rank 0 runs the trainer, then one broadcast carries [iteration, weights, normalisers]
    back (the only exchange on the policy side). Trajectories do not depend on the sharding, so the run equals train() on one process.
    overlap=True is train(overlap=True) across ranks: frame f+1 is relaunched on every rank before frame f's tuples are packed and gathered (the collective
    runs beside frame f+1), rank 0 trains on frame f-1's rows meanwhile, and the broadcast weights are parked for the next launch when the normalisers have not
    changed. Every tuple still reaches the trainer exactly once, in env-id order per frame; the policy a frame runs with is up to two frames staler.
    WHAT IS GUARANTEED about the tuple stream: with a send block that holds a frame's rows (the default block of N / 8 rows per rank once the lock-step start has
    spread out; always in the gloo tests' 32-env shards) the sequential mode equals train() in one process bit for bit and does not depend on the number of ranks.
    When a frame produces more rows than the block holds, the surplus is carried to later frames (never dropped): every tuple arrives exactly once and each env's
    tuples arrive in time order, but the interleaving across envs -- and with it the trainer's minibatches -- then depends on the block size and the number of
    ranks. Pass block_rows = the worst case (2 x envs per rank) when run-to-run equality across world sizes matters more than the 1.2 MB block. Both schedules end
    with a flush: gathers repeat until no rank's ring holds a carried row, so `tuples` == the rows the engines completed whatever the block size.
    mode="data_parallel": no trainer rank at all -- see train_data_parallel."""
    if mode == "data_parallel":      # no trainer rank: every rank trains on its own tuples, gradients all-reduced (train_data_parallel above)
        return train_data_parallel(arg_file, data_root, global_envs, dist, max_iters=max_iters, max_frames=max_frames, extra_args=extra_args, seed=seed,
                                   trainer_device=trainer_device, local_device_id=local_device_id, scenario_cls=scenario_cls, trainer_lib=trainer_lib, overlap=overlap)
    import torch
    from .sharding import ShardedRollout
    args = parse_arg_file(os.path.join(data_root, arg_file))
    args.update({k: str(v) for k, v in (extra_args or {}).items()})
    geti = lambda k, d: int(args.get(k, d)); getf = lambda k, d: float(args.get(k, d))
    rank = dist.get_rank()

    def make(n, off):
        ea = dict(extra_args or {}); ea["global_env_offset"] = off
        return scenario_cls(arg_file, n, data_root=data_root, device_id=local_device_id, extra_args=ea)
    sr = ShardedRollout(make, global_envs, dist=dist, device=device, pipelined=overlap, block_rows=block_rows)
    b = sr.batch
    hdr_dev = torch.device("cpu") if sr.staged else sr.device     # (a gloo group over device buffers: the loop's small headers stay on the host)
    t = None
    if rank == 0:
        solver = os.path.join(data_root, args["policy_solver"])
        m = re.search(r'net:\s*"([^"]+)"', open(solver).read())
        train_net = os.path.join(data_root, m.group(1)) if m else os.path.join(data_root, args["policy_net"].replace("_deploy", "_train"))
        tkw = dict(mem_size=geti("trainer_replay_mem_size", 500000), num_init_samples=geti("trainer_num_init_samples", 200),
                   steps_per_iter=geti("trainer_num_steps_per_iters", 1), freeze_target_iters=geti("trainer_freeze_target_iters", 0),
                   init_input_offset_scale=args.get("trainer_init_input_offset_scale", "false").lower() == "true", seed=seed, device=trainer_device)
        t = _make_trainer(args, data_root, train_net, solver, b, tkw, trainer, trainer_lib)
        t.SetOutputOffsetScale(*b.BuildNNOutputOffsetScale())
    exp = dict(rate=getf("exp_rate", 0.2), temp=getf("exp_temp", 0.025), base=getf("exp_base_rate", 0.002))
    init = dict(rate=getf("init_exp_rate", 1.0), temp=getf("init_exp_temp", 20.0), base=getf("init_exp_base_rate", 1.0))
    n_anneal, n_base_anneal, n_curr = geti("trainer_num_anneal_iters", 1), geti("exp_base_anneal_iters", 1), geti("trainer_curriculum_iters", 0)
    chunk = max(1, geti("tuple_buffer_size", 32))
    max_iters = max_iters if max_iters is not None else geti("trainer_max_iter", 10 ** 9)

    last_norm = [None]

    def sync(push):
        # header [push?, iteration, normalisers changed?] from the trainer rank, then (if push) the policy itself
        norm_new = 1
        if rank == 0 and push:
            norm = t.GetOffsetScale()
            norm_new = 0 if (last_norm[0] is not None and all(np.array_equal(a, c) for a, c in zip(norm, last_norm[0]))) else 1
            last_norm[0] = norm
        hdr = np.array([1 if push else 0, t.GetIter() if t is not None else 0, norm_new], np.int64)
        th = torch.from_numpy(hdr).to(hdr_dev)
        dist.broadcast(th, src=0)
        push, it, norm_new = (int(x) for x in th.tolist())
        if push:
            full = bool(norm_new) or not overlap      # (the sequential schedule keeps the one-call form: bit-identical to train())
            if rank == 0:
                sr.broadcast_policy(t.GetWeights(), *(last_norm[0] if full else (None,) * 4), src=0, normalizers=full, want_host=False)
            else:
                sr.broadcast_policy(src=0, normalizers=full, want_host=False)
        b.SetExplore(1, anneal(it, n_anneal, init["rate"], exp["rate"]), anneal(it, n_anneal, init["temp"], exp["temp"]), anneal(it, n_base_anneal, init["base"], exp["base"]))
        phase = 1.0 if n_curr < 1 else min(max(it / float(n_curr), 0.0), 1.0)
        b.SetTerrainParamsLerp(phase if it > 0 or n_curr < 1 else 0.0)
        return it

    it = sync(True)
    frames = tuples = 0
    t0 = time.time()

    ph = {"rollout": 0.0, "gather_end": 0.0, "to_host": 0.0, "add_tuples": 0.0, "train": 0.0, "gather_begin": 0.0, "policy_sync": 0.0}   # host wall-clock by phase, this rank
    clk = time.perf_counter

    def consume(g):
        """rank 0: gathered rows (tensors) -> the trainer, 32 at a time"""
        nonlocal tuples
        if rank != 0 or g is None:
            return False
        c0 = clk()
        rows, flags = g[0], g[1]
        if hasattr(rows, "cpu"):
            rows, flags = rows.cpu().numpy(), flags.cpu().numpy().astype(np.uint32)
        ph["to_host"] += clk() - c0
        feed_chunks(t, rows, flags, chunk, ph)
        tuples += len(rows)
        return len(rows) > 0

    import contextlib
    # the framework ops of the loop (header read-backs, concatenations, copies) on a stream of their own: the legacy default stream synchronises with every
    # blocking stream of the process, and the engine's CU-masked frame streams (-reserve_cus=) are such -- each op would wait for the frame in flight
    side_ctx = torch.cuda.stream(torch.cuda.Stream(device=sr.device)) if sr.on_gpu else contextlib.nullcontext()
    if not overlap:
        with side_ctx:
            while it < max_iters and (max_frames is None or frames < max_frames):
                sr.Update(1.0 / 30.0)
                frames += 1
                it = sync(consume(sr.gather_tuples(dst=0)))
            # flush: rows a small block carried over (bursts above block_rows) still sit in some rank's ring; every rank takes part in every gather
            got = False
            while True:
                tl = torch.tensor([b.TupleStats()["pending"]], dtype=torch.int64, device=hdr_dev)
                dist.all_reduce(tl)
                if int(tl.item()) == 0:
                    break
                got = consume(sr.gather_tuples(dst=0)) or got
            it = sync(got)
    else:
        with side_ctx:
            sr.UpdateBegin(1.0 / 30.0)
            more = True
            while more:
                frames += 1
                more = it < max_iters and (max_frames is None or frames < max_frames)
                c0 = clk()
                if more:
                    sr.UpdateEndBegin(1.0 / 30.0)            # frame f ended, frame f+1 runs ...
                else:
                    sr.UpdateEnd()
                c1 = clk()
                g = sr.gather_tuples_end(dst=0) if sr._pending is not None else None
                c2 = clk()
                got = consume(g)                             # ... rank 0 trains on frame f-1's rows (gathered during frame f),
                c3 = clk()
                sr.gather_tuples_begin(dst=0)                # frame f's rows are packed and put on the wire beside frame f+1,
                c4 = clk()
                it = sync(got)                               # and the new weights are parked for the next launch
                ph["rollout"] += c1 - c0; ph["gather_end"] += c2 - c1; ph["gather_begin"] += c4 - c3; ph["policy_sync"] += clk() - c4
            # flush: the last frame's gather, then whatever small blocks carried over (every rank takes part in every gather)
            left = 1
            while left > 0:
                consume(sr.gather_tuples_end(dst=0))
                st = b.TupleStats()
                tl = torch.tensor([st["pending"]], dtype=torch.int64, device=hdr_dev)
                dist.all_reduce(tl)
                left = int(tl.item())
                if left > 0:
                    sr.gather_tuples_begin(dst=0)
            try:
                b.SetTuplePipelining(False)
            except Exception:
                pass       # rows a small block carried over sit in the idle ring: the run ends with them undelivered, as the reference's ends with part-filled scene buffers
            it = sync(rank == 0 and t.GetIter() > 0)
    dt = time.time() - t0
    out = dict(frames=frames, iters=it, seconds=dt, env_steps_per_s=frames * 20.0 * global_envs / dt, rank=rank, batch=b, phases=ph)
    if rank == 0:
        out.update(tuples=tuples, weights=t.GetWeights(), offset_scale=t.GetOffsetScale(), carried_rows=sr.carried_rows)
    return out
