"""The C-ABI library loads on a CPU-only box and exports every symbol include/dtrl.h declares; the product has no CPU fallback."""
import ctypes
import os
import re

import pytest

from conftest import REPO, HIP_LIB, EMUL_LIB, REFDATA


def header_symbols():
    txt = open(os.path.join(REPO, "include", "dtrl.h")).read()
    return sorted(set(re.findall(r"\b(dtrl_[a-z_0-9]+)\s*\(", txt)))


def test_header_lists_expected_entry_points(da):
    syms = header_symbols()
    assert "dtrl_create" in syms and "dtrl_step" in syms and "dtrl_drain_tuples" in syms
    assert sorted(da.ABI_SYMBOLS) == syms


@pytest.mark.parametrize("path", [HIP_LIB, EMUL_LIB])
def test_library_exports_every_declared_symbol(path):
    lib = ctypes.CDLL(path)
    for s in header_symbols():
        assert hasattr(lib, s), s


def test_every_entry_point_cites_the_reference():
    txt = open(os.path.join(REPO, "include", "dtrl.h")).read()
    assert txt.count("Replaces:") >= 12
    assert re.search(r"scenarios/ScenarioExp\.cpp:\d+", txt) and re.search(r"sim/SimCharacter\.cpp:\d+", txt)


def test_no_cpu_fallback_when_device_missing(da):
    """Without a HIP device dtrl_create must fail loudly (DTRL_ERR_NO_DEVICE); on a GPU box it must succeed."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = os.path.exists("/dev/kfd")
    if has_gpu:
        b = da.BatchScenario("args/sim_dog_args.txt", 1, data_root=REFDATA)
        assert b.D == 23
    else:
        with pytest.raises(da.DtrlError) as ei:
            da.BatchScenario("args/sim_dog_args.txt", 1, data_root=REFDATA)
        assert "(3)" in str(ei.value) and "no CPU fallback" in str(ei.value)


def test_product_does_not_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/ (and nothing reads /root/reference at run time)."""
    pkg = os.path.join(REPO, "deepterrainrl_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip")):
                src = open(os.path.join(root, f)).read()
                assert "oracle/" not in src.replace("oracle/or_", "ORC_DOC").replace("oracle/model.py", "ORC_DOC") or f.endswith((".h", ".cpp", ".hip")), f
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libdtrl_oracle" not in src, f
                assert "libdtrl_emul" not in src and "_lib_path" not in src, f
    assert not os.path.exists(os.path.join(pkg, "lib", "libdtrl_emul.so")) and not os.path.isdir(os.path.join(pkg, "csrc", "emul")), "the lane-loop test backend must live under tests/, not in the product package"


def test_hardware_queue_count_is_opt_in():
    """The engine's env-group streams must not share a HIP hardware queue. Importing the package leaves the process environment alone
    (ADVICE r2); deepterrainrl_amd.configure_hw_queues() sets GPU_MAX_HW_QUEUES=8 when the user has not set it and never overrides a user's value."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); import deepterrainrl_amd as da; a = os.environ.get('GPU_MAX_HW_QUEUES'); r = da.configure_hw_queues(); "
            "print(a, r, os.environ['GPU_MAX_HW_QUEUES'])" % repo)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.check_output([sys.executable, "-c", code], env=env).decode().split() == ["None", "8", "8"]
    env["GPU_MAX_HW_QUEUES"] = "5"
    assert subprocess.check_output([sys.executable, "-c", code], env=env).decode().split() == ["5", "5", "5"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher must start two ranks (--gpus used to be ignored). --dry-launch makes every rank print
    its placement and exit, so the launch path is checked on a box without a GPU: distinct RANK / LOCAL_RANK / device / global env offset, world = 2,
    the host worker threads divided between the ranks; and a plain --gpus 1 stays in-process."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DTRL_HOST_THREADS")}
    out = subprocess.check_output([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-launch", "--envs-per-gpu", "96"], env=env, stderr=subprocess.STDOUT, timeout=600).decode()
    recs = [json.loads(l) for l in out.splitlines() if l.startswith("{") and "dry_launch" in l]
    assert len(recs) == 2, out
    recs.sort(key=lambda r: r["rank"])
    assert [r["rank"] for r in recs] == [0, 1] and [r["local_rank"] for r in recs] == [0, 1] and all(r["world"] == 2 for r in recs)
    assert [r["global_env_offset"] for r in recs] == [0, 96] and [r["device"] for r in recs] == ["cuda:0", "cuda:1"]
    one = subprocess.check_output([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--dry-launch"], env=env, stderr=subprocess.STDOUT, timeout=600).decode()
    r1 = [json.loads(l) for l in one.splitlines() if l.startswith("{") and "dry_launch" in l]
    assert len(r1) == 1 and r1[0]["world"] == 1 and r1[0]["envs_per_gpu"] == 4096
    assert recs[0]["host_threads"] == max(1, min(16, (os.cpu_count() or 2) // 4)) and r1[0]["host_threads"] == max(1, min(16, (os.cpu_count() or 2) // 2))


def test_bench_starts_eight_ranks_the_way_the_driver_does():
    """The 8-GPU shape of the scaling run, twice: `python bench.py --gpus 8` on its own (self-launch) and under the driver's launcher line
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...`): eight ranks, local ranks 0..7 on
    devices cuda:0..7, global env offsets 0, 4096, ..., 28672 (BASELINE configs[3]: 32768 envs sharded over 8 GPUs), the node's host worker threads divided by eight."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DTRL_HOST_THREADS")}
    bench = os.path.join(repo, "bench.py")
    for cmd in ([sys.executable, bench, "--gpus", "8", "--dry-launch"],
                [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29671", bench, "--gpus", "8", "--steps", "5", "--warmup", "2", "--dry-launch"]):
        out = subprocess.check_output(cmd, env=env, stderr=subprocess.STDOUT, timeout=900).decode()
        recs = sorted((json.loads(l) for l in out.splitlines() if l.startswith("{") and "dry_launch" in l), key=lambda r: r["rank"])
        assert len(recs) == 8, out
        assert [r["rank"] for r in recs] == list(range(8)) and [r["local_rank"] for r in recs] == list(range(8)) and all(r["world"] == 8 and r["gpus_arg"] == 8 for r in recs)
        assert [r["global_env_offset"] for r in recs] == [4096 * k for k in range(8)] and [r["device"] for r in recs] == ["cuda:%d" % k for k in range(8)]
        assert len({r["host_threads"] for r in recs}) == 1 and recs[0]["host_threads"] >= 1


def test_bench_eight_ranks_end_to_end_on_the_check_build_over_gloo():
    """VERDICT r5 #6a: the first 8-GPU run must be boring. `bench.py --gpus 8` END TO END (not --dry-launch) on a box without GPUs: eight ranks of the lane-loop check build
    (--backend emul-tests-only, tests/emul/libdtrl_emul.so) over a gloo group walk the whole protocol -- self-launch through torch.distributed.run, env-id sharding with
    global offsets, the pre-roll agreement, barrier + max-over-ranks timing of exactly --steps frames, the exchange leg (pipelined tuple gather to rank 0 + packed policy
    broadcast, configs[3] shape) -- and rank 0 prints ONE line: n_gpus 8, a `rccl`-shaped record with 8 ranks, scaling weak, value = all ranks' env-steps over the slowest
    rank's time, every exploration tuple exactly once. The line is marked NOT A MEASUREMENT; on the GPU node the same code runs on libdtrl.so over RCCL."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DTRL_HOST_THREADS")}
    env["DTRL_TESTS_ONLY_EMUL"] = "1"
    bench = os.path.join(repo, "bench.py")
    # without the switch the check build is refused (it is not a fallback)
    refused = subprocess.run([sys.executable, bench, "--backend", "emul-tests-only", "--dry-launch"], env={k: v for k, v in env.items() if k != "DTRL_TESTS_ONLY_EMUL"},
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert refused.returncode != 0 and b"DTRL_TESTS_ONLY_EMUL" in refused.stdout
    n, steps = 6, 3
    out = subprocess.check_output([sys.executable, bench, "--gpus", "8", "--backend", "emul-tests-only", "--envs-per-gpu", str(n), "--steps", str(steps), "--warmup", "1", "--repeats", "2",
                                   "--preroll-max", "20", "--exchange-steps", "45", "--bcast-every", "10", "--no-cpu-baseline"], env=env, stderr=subprocess.STDOUT, timeout=1500).decode()
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-3000:]                   # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["higher_is_better"] is True and d["steps"] == steps and d["repeats"] == 2
    assert d["rccl"]["ranks"] == 8 and d["rccl"]["backend"] == "gloo" and "NOT A MEASUREMENT" in d["data"] and d["backend"].startswith("emul-tests-only")
    assert d["config"]["global_envs"] == 8 * n and d["config"]["env_steps_per_step"] == 8 * n * 20 and "x8" in d["config"]["parallelism"]
    # value = the units ALL ranks processed / the (max over ranks) time of the median window
    assert abs(d["value"] - 8 * n * steps * 20 / d["window_s"]["median"]) < 1e-6 * d["value"]
    assert abs(d["ms_per_step"] - d["window_s"]["median"] / steps * 1e3) < 1e-9
    ex = d["exchange"]
    assert "error" not in ex and ex["steps"] == 45 and ex["dropped_tuples"] == 0 and "8 GPUs" in ex["workload"] and "gather to rank 0" in ex["collective"]
    assert ex["collective_bytes_per_frame"]["received_by_rank0"] == 8 * ex["collective_bytes_per_frame"]["sent_per_rank"] > 0
    assert ex["tuples"] > 0                                # 48 envs x 45 frames past the two warm-up cycles: tuples from every shard reached rank 0's replay ring
