"""N > 1 path on CPU: world_size-2 gloo run of the sharded rollout (lane-loop test backend) -- shard-invariant trajectories,
tuple gather to the trainer rank and policy broadcast equal a single-process run over the same global env ids."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, REFDATA, EmulScenario

Scenario = EmulScenario

WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import torch, torch.distributed as dist
from conftest import REFDATA, EmulScenario, dog_policy
Scenario = EmulScenario
from oracle import model as om
import deepterrainrl_amd as da
from deepterrainrl_amd.sharding import ShardedRollout
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
def make(n, off):
    return Scenario("args/opt_args_train_mace.txt", n, data_root=REFDATA, extra_args=dict(terrain_seed=300, rand_seed=9, global_env_offset=off))
sr = ShardedRollout(make, 4, dist=dist, block_rows={block_rows!r})
assert sr.cap == ({block_rows!r} or 64)
pol = dog_policy(om)
if rank == 0:
    sr.broadcast_policy(pol[1], pol[2], pol[3], pol[4], pol[5], src=0)
else:
    sr.broadcast_policy(src=0)
rows, flags, ids = [], [], []
for f in range(70):
    sr.Update()
    g = sr.gather_tuples(dst=0)
    if rank == 0:
        rows.append(g[0]); flags.append(g[1]); ids.append(g[2])
for f in range(12):                      # no stepping: rows a small block carried over are flushed (every rank takes part in every gather)
    g = sr.gather_tuples(dst=0)
    if rank == 0:
        rows.append(g[0]); flags.append(g[1]); ids.append(g[2])
assert sr.batch.TupleStats()["pending"] == 0 and sr.batch.TupleStats()["dropped"] == 0
q, qd = sr.batch.PoseVel()
np.save(os.path.join({out!r}, "q_rank%d.npy" % rank), q)
if rank == 0:
    np.savez(os.path.join({out!r}, "tuples.npz"), rows=np.concatenate(rows), flags=np.concatenate(flags), ids=np.concatenate(ids))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("block_rows", [None, 1])
def test_two_rank_gloo_matches_single_process(tmp_path, da, om, block_rows):
    """block_rows=1: a send block far smaller than a frame's rows -- the gather to rank 0 delivers one row per rank and frame and the engine carries
    the rest to later frames; the trainer still receives every tuple, each env's in time order."""
    from conftest import dog_policy
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(repo=REPO, out=str(tmp_path), block_rows=block_rows))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(29611 + (block_rows or 0) * 7), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # single process over the same 4 global envs
    pol = dog_policy(om)
    b = Scenario("args/opt_args_train_mace.txt", 4, data_root=REFDATA, extra_args=dict(terrain_seed=300, rand_seed=9))
    b.SetPolicy(pol[1], *pol[2:])
    rows, flags, ids = [], [], []
    for f in range(70):
        b.Update()
        rr, ff, ii = b.DrainTuples()
        rows.append(rr); flags.append(ff); ids.append(ii)
    rows = np.concatenate(rows); flags = np.concatenate(flags); ids = np.concatenate(ids)
    q, _ = b.PoseVel()
    q0 = np.load(tmp_path / "q_rank0.npy"); q1 = np.load(tmp_path / "q_rank1.npy")
    assert np.array_equal(q[:2], q0) and np.array_equal(q[2:], q1)              # shard-invariant trajectories
    t = np.load(tmp_path / "tuples.npz")
    assert len(t["rows"]) == len(rows) and len(rows) >= 8
    for e in range(4):                                                            # same tuples per global env, same order
        assert np.array_equal(t["rows"][t["ids"] == e], rows[ids == e]) and np.array_equal(t["flags"][t["ids"] == e], flags[ids == e])


TRAIN_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import torch, torch.distributed as dist
from conftest import REFDATA, EmulScenario
from deepterrainrl_amd import train_loop
dist.init_process_group(backend="gloo")
st = train_loop.train_distributed("args/opt_args_train_mace.txt", REFDATA, 64, dist, max_frames={frames}, trainer_device="cpu", scenario_cls=EmulScenario, extra_args={extra!r}, trainer={trainer!r}, trainer_lib={lib!r})
if dist.get_rank() == 0:
    np.savez(os.path.join({out!r}, "dist_train.npz"), weights=st["weights"], iters=st["iters"], tuples=st["tuples"], in_off=st["offset_scale"][0])
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("trainer,frames", [("torch", 60), ("hip", 45)])
def test_two_rank_training_equals_single_process(tmp_path, da, trainer, frames):
    """Rollout shards + tuple gather + trainer on rank 0 + policy broadcast (gloo, world size 2) == train() in one process, bit for bit -- with the PyTorch
    peer trainer and with the native trainer step (its plain-loop check build here; rows reach it through the staging area on both sides)."""
    lib = os.path.join(REPO, "tests", "emul", "libdtrl_trainer_emul.so") if trainer == "hip" else None
    # (identity input normaliser: estimated from 30 tuples instead of the file's 50 000, a near-constant terrain feature gets a scale of 1e9 and the float32
    # net overflows on the first batch -- cNeuralNet::CalcOffsetScale has no floor either; the normaliser itself is covered by tests/test_trainer.py)
    extra = {"terrain_seed": 3, "trainer_num_init_samples": 30, "trainer_replay_mem_size": 512, "trainer_freeze_target_iters": 4,
             "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"}
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER.format(repo=REPO, out=str(tmp_path), extra=extra, trainer=trainer, lib=lib, frames=frames))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29613" if trainer == "torch" else "29615", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(tmp_path / "dist_train.npz")
    # the same run in ONE process (also with a single OpenMP thread: float32 GEMM/conv reductions depend on the thread count)
    single = tmp_path / "train_single.py"
    single.write_text("import os, sys, numpy as np\nsys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
                      "from conftest import REFDATA, EmulScenario\nfrom deepterrainrl_amd import train_loop\n"
                      "st = train_loop.train('args/opt_args_train_mace.txt', REFDATA, num_envs=64, max_frames=%d, trainer_device='cpu', scenario_cls=EmulScenario, extra_args=%r, trainer=%r, trainer_lib=%r)\n"
                      "np.savez(os.path.join(%r, 'single_train.npz'), weights=st['weights'], iters=st['iters'], tuples=st['tuples'], in_off=st['offset_scale'][0])\n"
                      % (REPO, REPO, frames, extra, trainer, lib, str(tmp_path)))
    r = subprocess.run([sys.executable, str(single)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    s1 = np.load(tmp_path / "single_train.npz")
    assert int(d["iters"]) == int(s1["iters"]) >= 1 and int(d["tuples"]) == int(s1["tuples"])
    assert np.array_equal(d["weights"], s1["weights"]) and np.array_equal(d["in_off"], s1["in_off"]) and np.all(np.isfinite(d["weights"]))


OVERLAP_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import torch, torch.distributed as dist
from conftest import REFDATA, EmulScenario
from deepterrainrl_amd import train_loop
dist.init_process_group(backend="gloo")
st = train_loop.train_distributed("args/opt_args_train_mace.txt", REFDATA, 64, dist, max_frames=60, trainer_device="cpu", scenario_cls=EmulScenario, extra_args={extra!r},
                                  trainer="hip", trainer_lib={lib!r}, overlap={overlap!r}, block_rows={block_rows!r})
drained = torch.tensor([st["batch"].TupleStats()["drained"], st["batch"].TupleStats()["dropped"], st["batch"].TupleStats()["pending"]], dtype=torch.int64)
dist.all_reduce(drained)
if dist.get_rank() == 0:
    np.savez(os.path.join({out!r}, "overlap_train.npz"), weights=st["weights"], iters=st["iters"], tuples=st["tuples"], frames=st["frames"], drained=drained.numpy(),
             carried=st["carried_rows"])
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("overlap,block_rows", [(True, None), (False, 2)], ids=["overlapped", "sequential_small_block"])
def test_two_rank_training_delivers_every_tuple(tmp_path, da, overlap, block_rows):
    """train_distributed on two gloo ranks with the native trainer's check build. overlapped: frames relaunched before the gather, rank 0 training on the
    previous frame's rows, weights parked for the next launch. sequential_small_block: a send block of 2 rows per rank, far below the lock-step start's bursts
    (32 envs per rank finish their first cycles within a few frames of each other) -- rows are carried from frame to frame and the run ends with the flush.
    What is guaranteed either way: every tuple the ranks' engines completed reaches the trainer exactly once, none dropped, none left in a ring; the trainer
    iterates and the weights stay finite."""
    lib = os.path.join(REPO, "tests", "emul", "libdtrl_trainer_emul.so")
    extra = {"terrain_seed": 3, "trainer_num_init_samples": 30, "trainer_replay_mem_size": 512, "trainer_freeze_target_iters": 4,
             "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"}
    script = tmp_path / "overlap_worker.py"
    script.write_text(OVERLAP_WORKER.format(repo=REPO, out=str(tmp_path), extra=extra, lib=lib, overlap=overlap, block_rows=block_rows))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29617" if overlap else "29619", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(tmp_path / "overlap_train.npz")
    drained, dropped, pending = (int(x) for x in d["drained"])
    assert int(d["frames"]) == 60 and int(d["iters"]) >= 1 and np.all(np.isfinite(d["weights"]))
    assert int(d["tuples"]) == drained >= 40 and dropped == 0 and pending == 0
    if block_rows:
        assert int(d["carried"]) > 0          # the small block did carry rows (otherwise this case checks nothing)


def test_shard_range():
    from deepterrainrl_amd.sharding import shard_range
    assert [shard_range(32768, 8, r) for r in (0, 7)] == [(0, 4096), (28672, 4096)]
    assert [shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]


def test_overlapped_exchange_equals_the_synchronous_one(om):
    """gather_tuples_begin / UpdateBegin / gather_tuples_end / UpdateEnd (the bench's exchange leg: the collective overlaps the next frame) hands the
    trainer the same rows, in the same order, as the synchronous Update + gather_tuples loop; one packed policy broadcast installs the same policy."""
    import torch
    from deepterrainrl_amd.sharding import ShardedRollout
    pol = dog_policy_()
    def make(n, off):
        return Scenario("args/opt_args_train_mace.txt", n, data_root=REFDATA, extra_args=dict(terrain_seed=300, rand_seed=9, global_env_offset=off))
    a = ShardedRollout(make, 6); b = ShardedRollout(make, 6)
    a.broadcast_policy(pol[1], *pol[2:]); b.broadcast_policy(pol[1], *pol[2:])
    rows_a, rows_b = [], []
    b.UpdateBegin()
    for f in range(80):
        a.Update()
        g = a.gather_tuples()
        rows_a.append(np.concatenate([g[0], g[1][:, None].astype(np.float32), g[2][:, None].astype(np.float32)], axis=1))
        b.UpdateEnd(); b.gather_tuples_begin(); b.UpdateBegin()
        r, fl, ids = b.gather_tuples_end()
        rows_b.append(np.concatenate([r.numpy(), fl.numpy()[:, None].astype(np.float32), ids.numpy()[:, None].astype(np.float32)], axis=1))
    b.UpdateEnd()
    ra, rb = np.concatenate(rows_a), np.concatenate(rows_b)
    assert len(ra) >= 6 and np.array_equal(ra, rb)
    a.Update()   # b is one frame ahead (its frame 81 has run)
    assert np.array_equal(a.batch.PoseVel()[0], b.batch.PoseVel()[0])


def dog_policy_():
    from conftest import dog_policy
    from oracle import model as om
    return dog_policy(om)


DP_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import torch, torch.distributed as dist
from conftest import REFDATA, EmulScenario
from deepterrainrl_amd import train_loop
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
st = train_loop.train_distributed("args/opt_args_train_mace.txt", REFDATA, 128, dist, max_frames=80, trainer_device="cpu", scenario_cls=EmulScenario, extra_args={extra!r},
                                  trainer="hip", trainer_lib={lib!r}, mode="data_parallel")
t = st["trainer"]
X = t.mem[:t.num_tuples, 1:1 + t.S].to(torch.float64)
stats = torch.cat([torch.tensor([float(X.shape[0])], dtype=torch.float64), X.sum(0), (X * X).sum(0)])
np.savez(os.path.join({out!r}, "dp_rank%d.npz" % rank), weights=st["weights"], iters=st["iters"], actor_iters=st["actor_iters"], tuples=st["tuples"], frames=st["frames"],
         in_off=st["offset_scale"][0], in_scale=st["offset_scale"][1], hist=t.nt.get_params(2), stats=stats.numpy(), drained=st["batch"].TupleStats()["drained"])
dist.barrier(); dist.destroy_process_group()
'''


def test_two_rank_data_parallel_training(tmp_path, da):
    """train_distributed(mode="data_parallel") on two gloo ranks (native trainer's check build): no tuple gather and no weight broadcast, yet both ranks end with
    bit-identical weights and solver history, the same iteration counters, a normaliser pooled over BOTH ranks' begin states; every rank trained on its own tuples
    only (different replay contents), the weights moved and stayed finite."""
    lib = os.path.join(REPO, "tests", "emul", "libdtrl_trainer_emul.so")
    extra = {"terrain_seed": 3, "trainer_num_init_samples": 60, "trainer_replay_mem_size": 512, "trainer_freeze_target_iters": 4,
             "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"}
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER.format(repo=REPO, out=str(tmp_path), extra=extra, lib=lib))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29619", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(tmp_path / "dp_rank0.npz"), np.load(tmp_path / "dp_rank1.npz")
    assert int(a["frames"]) == int(b["frames"]) == 80
    assert int(a["iters"]) == int(b["iters"]) >= 2 and int(a["actor_iters"]) == int(b["actor_iters"])
    assert np.array_equal(a["weights"], b["weights"]) and np.array_equal(a["hist"], b["hist"]) and np.all(np.isfinite(a["weights"]))
    assert np.abs(a["hist"]).max() > 0                                                   # updates happened
    assert int(a["tuples"]) == int(a["drained"]) >= 10 and int(b["tuples"]) == int(b["drained"]) >= 10     # each rank kept exactly what its own engine produced
    assert not np.array_equal(a["stats"], b["stats"])                                    # ... and the two replay memories hold different experience


# ---- the 8-rank shape: the node the path is written for has 8 GPUs and the builder has none with more than one, so the first real 8-GPU run must
# ---- not also be the first 8-rank run. gloo, lane-loop backend, small shards (the collectives, block sizing, carry-over, the remainder split and the gather list of
# ---- eight are what is under test, not the kernels)
WORKER8 = r'''
import os, sys, zlib, numpy as np
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import torch, torch.distributed as dist
from conftest import REFDATA, EmulScenario, dog_policy
from oracle import model as om
from deepterrainrl_amd.sharding import ShardedRollout, shard_range, default_block_rows
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
G = {global_envs}
def make(n, off):
    assert (off, n) == shard_range(G, world, rank)
    return EmulScenario("args/opt_args_train_mace.txt", n, data_root=REFDATA, extra_args=dict(terrain_seed=300, rand_seed=9, global_env_offset=off))
sr = ShardedRollout(make, G, dist=dist, block_rows={block_rows!r})
assert sr.cap == ({block_rows!r} or default_block_rows(G, world))
pol = dog_policy(om)
if rank == 0:
    sr.broadcast_policy(pol[1], pol[2], pol[3], pol[4], pol[5], src=0)
else:
    sr.broadcast_policy(src=0)
rows, flags, ids = [], [], []
for f in range({frames}):
    sr.Update()
    g = sr.gather_tuples(dst=0)
    if rank == 0:
        rows.append(g[0]); flags.append(g[1]); ids.append(g[2])
for f in range(40):                      # flush what small blocks carried over
    g = sr.gather_tuples(dst=0)
    if rank == 0:
        rows.append(g[0]); flags.append(g[1]); ids.append(g[2])
ts = sr.batch.TupleStats()
assert ts["pending"] == 0 and ts["dropped"] == 0
q, qd = sr.batch.PoseVel()
np.savez(os.path.join({out!r}, "w%d_rank%d.npz" % (world, rank)), q=q, off=shard_range(G, world, rank)[0], drained=ts["drained"], pol=sr.batch.PolicyOutput())
if rank == 0:
    np.savez(os.path.join({out!r}, "w%d_tuples.npz" % world), rows=np.concatenate(rows), flags=np.concatenate(flags), ids=np.concatenate(ids))
dist.barrier(); dist.destroy_process_group()
'''


def _launch(script, nproc, port, timeout=1500):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc, "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("block_rows", [None, 1], ids=["default_block", "one_row_block"])
def test_eight_rank_gather_is_independent_of_the_world_size(tmp_path, da, om, block_rows):
    """19 global envs on 8 ranks (shards of 3, 3, 3, 2, 2, 2, 2, 2: shard_range with a remainder), on 2 ranks and in one process: the same trajectories per global env,
    every tuple delivered to rank 0 exactly once, each env's tuples in time order and bit-identical across the three world sizes -- with the default send block
    (default_block_rows) and with a one-row block (seven of eight ranks carry rows from frame to frame; the run ends with the flush). The broadcast policy is the same
    on every rank: all shards ride the single-process trajectories, and the nets' last outputs (dtrl_get_policy_output) are those of the single process."""
    from conftest import dog_policy
    G, frames = 19, 40
    for world, port in ((8, 29631 + (block_rows or 0) * 10), (2, 29633 + (block_rows or 0) * 10)):
        script = tmp_path / ("worker_w%d.py" % world)
        script.write_text(WORKER8.format(repo=REPO, out=str(tmp_path), block_rows=block_rows, global_envs=G, frames=frames))
        _launch(script, world, port)
    pol = dog_policy(om)
    b = Scenario("args/opt_args_train_mace.txt", G, data_root=REFDATA, extra_args=dict(terrain_seed=300, rand_seed=9))
    b.SetPolicy(pol[1], *pol[2:])
    rows, flags, ids = [], [], []
    for f in range(frames):
        b.Update()
        rr, ff, ii = b.DrainTuples()
        rows.append(rr); flags.append(ff); ids.append(ii)
    rows = np.concatenate(rows); flags = np.concatenate(flags); ids = np.concatenate(ids)
    q, _ = b.PoseVel(); po = b.PolicyOutput()
    assert len(rows) >= 19
    for world in (8, 2):
        drained = 0
        for r in range(world):
            d = np.load(tmp_path / ("w%d_rank%d.npz" % (world, r)))
            off, n = int(d["off"]), len(d["q"])
            assert np.array_equal(d["q"], q[off:off + n]) and np.array_equal(d["pol"], po[off:off + n]), (world, r)
            drained += int(d["drained"])
        t = np.load(tmp_path / ("w%d_tuples.npz" % world))
        assert drained == len(t["rows"]) == len(rows), (world, drained, len(t["rows"]), len(rows))       # every tuple exactly once
        for e in range(G):
            assert np.array_equal(t["rows"][t["ids"] == e], rows[ids == e]) and np.array_equal(t["flags"][t["ids"] == e], flags[ids == e]), (world, e)


DP8_WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import torch, torch.distributed as dist
from conftest import REFDATA, EmulScenario
from deepterrainrl_amd import train_loop
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
st = train_loop.train_distributed("args/opt_args_train_mace.txt", REFDATA, {global_envs}, dist, max_frames={frames}, trainer_device="cpu", scenario_cls=EmulScenario, extra_args={extra!r},
                                  trainer="hip", trainer_lib={lib!r}, mode="data_parallel")
t = st["trainer"]
X = t.mem[:t.num_tuples, 1:1 + t.S].to(torch.float64)
stats = torch.cat([torch.tensor([float(X.shape[0])], dtype=torch.float64), X.sum(0), (X * X).sum(0)])
t.UpdateOffsetScale()          # the pooled normaliser (one all-reduce of count / sum / sum of squares): every rank calls it, after the training it did not feed
off, sc = t.GetOffsetScale()[:2]
np.savez(os.path.join({out!r}, "dp8_rank%d.npz" % rank), weights=st["weights"], iters=st["iters"], actor_iters=st["actor_iters"], tuples=st["tuples"], frames=st["frames"],
         hist=t.nt.get_params(2), stats=stats.numpy(), drained=st["batch"].TupleStats()["drained"], pooled_off=np.asarray(off), pooled_scale=np.asarray(sc))
dist.barrier(); dist.destroy_process_group()
'''


def test_eight_rank_data_parallel_training(tmp_path, da):
    """train_distributed(mode="data_parallel") on EIGHT gloo ranks (native trainer's check build; 131 global envs = shards of 17, 17, 17, 16, 16, 16, 16, 16: a rank's replay memory must hold a minibatch of 32 before its trainer steps): the per-frame
    agreement on the number of Train() calls, the two gradient all-reduces per call and the pooled input normaliser run with eight participants; all ranks end with
    bit-identical weights and solver history and the same counters, each kept exactly the tuples its own engine produced, and the normaliser every rank computes is
    the one of the POOLED begin states (HipMACETrainerDP.UpdateOffsetScale was never exercised)."""
    lib = os.path.join(REPO, "tests", "emul", "libdtrl_trainer_emul.so")
    extra = {"terrain_seed": 3, "trainer_num_init_samples": 48, "trainer_replay_mem_size": 256, "trainer_freeze_target_iters": 2, "tuple_buffer_size": 4,
             "init_exp_rate": 0.3, "init_exp_base_rate": 0.1, "trainer_init_input_offset_scale": "false"}
    script = tmp_path / "dp8_worker.py"
    script.write_text(DP8_WORKER.format(repo=REPO, out=str(tmp_path), extra=extra, lib=lib, global_envs=131, frames=120))
    _launch(script, 8, 29651, timeout=2400)
    R = [np.load(tmp_path / ("dp8_rank%d.npz" % r)) for r in range(8)]
    a = R[0]
    assert int(a["frames"]) == 120 and int(a["iters"]) >= 2 and np.abs(a["hist"]).max() > 0 and np.all(np.isfinite(a["weights"]))
    pooled = sum(r["stats"] for r in R)
    S = (len(pooled) - 1) // 2
    n = pooled[0]; mean = pooled[1:1 + S] / n
    std = np.sqrt(np.maximum(pooled[1 + S:] / n - mean * mean, 0.0))
    exp_scale = np.where(std == 0, 0.0, 1.0 / np.where(std == 0, 1.0, std))
    for k, r in enumerate(R):
        assert int(r["iters"]) == int(a["iters"]) and int(r["actor_iters"]) == int(a["actor_iters"]) and int(r["frames"]) == 120, k
        assert np.array_equal(r["weights"], a["weights"]) and np.array_equal(r["hist"], a["hist"]), k
        assert int(r["tuples"]) == int(r["drained"]) >= 1, k
        assert np.allclose(r["pooled_off"], -mean, rtol=1e-6, atol=1e-9) and np.allclose(r["pooled_scale"], exp_scale, rtol=1e-5, atol=1e-9), k
    assert len({float(r["stats"][0]) for r in R} | {tuple(r["stats"][1:4]) for r in R}) > 2        # the replay memories hold different experience
