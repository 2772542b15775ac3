"""N > 1 path on CPU: world_size-2 (and 3, uneven shards) gloo jobs shard a batch of groups, decode their
shard and all-reduce the four error counters; the result must equal the single-process counters."""
import json
import os
import socket
import subprocess
import sys

import pytest

import oracle_abi as oa

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_groups_covers_every_group_once():
    import importlib.util
    spec = importlib.util.spec_from_file_location("lnsfaid_dist", os.path.join(oa.PKG_DIR, "dist.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for n in (0, 1, 5, 32, 2048):
        for world in (1, 2, 3, 8):
            ranges = [m.shard_groups(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            assert max(b - a for a, b in ranges) - min(b - a for a, b in ranges) <= 1
    with pytest.raises(ValueError):
        m.shard_groups(4, 2, 2)
    assert m.allreduce_counters([1, 2, 3, 4]) == [1, 2, 3, 4]


@pytest.mark.parametrize("world", [2, 3])
def test_counters_allreduce_over_gloo(abi, code50, tmp_path, world):
    n_groups, eb_n0 = 4, 3.5
    out = tmp_path / "total.json"
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(rank), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, WORKER, str(n_groups), str(eb_n0), str(out)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = json.load(open(out))
    cfg = abi.default_cfg(2, 10)
    fix = oa.ReferenceChannel(code50, 101, 13.0).groups(eb_n0, n_groups)
    dec, _ = oa.decode_mt(code50, cfg, fix, n_groups)
    want = oa.Oracle(code50, cfg).count_errors(dec, None, n_groups)
    assert got["world"] == world and got["total"] == want


BENCH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def _bench_line(argv, timeout=900):
    """Run bench.py plainly (the way the driver does) and parse the one JSON line it prints."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, BENCH] + argv, env=env, stdout=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_launcher_spawns_one_worker_per_rank():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two fresh workers (torch.distributed.run),
    the weak leg gives every rank its own groups, the strong leg splits the groups, counters are all-reduced (gloo here;
    the CPU port stands in for the GPU library: this checks the plumbing, not the kernels)."""
    d = _bench_line(["--gpus", "2", "--launcher-selftest", "--groups", "4", "--steps", "1", "--warmup", "0", "--eb-n0", "3.5"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["invalid_for_measurement"] is True
    cfg = d["config"]
    assert cfg["world_size"] == 2 and len(cfg["per_rank_counters"]) == 2
    tot = cfg["counters_TestFrame_ErrorFrame_ErrorBits_LT3"]
    assert tot == [sum(r[i] for r in cfg["per_rank_counters"]) for i in range(4)]
    assert tot[0] == 2 * 4 * 32                    # weak: 4 groups on each of the 2 ranks
    st = d["strong"]
    assert st["total_groups"] == 4 and st["groups_on_rank0"] == 2
    assert st["counters_TestFrame_ErrorFrame_ErrorBits_LT3"][0] == 4 * 32  # strong: 4 groups in total
    assert st["counters_TestFrame_ErrorFrame_ErrorBits_LT3"] == [sum(r[i] for r in st["per_rank_counters"]) for i in range(4)]


@pytest.mark.gpu
def test_bench_two_ranks_through_the_hip_library():
    """Two ranks through liblnsfaid.so, launched by bench.py itself.  With two GPUs: one each, RCCL.  On a one-GPU box both
    ranks share the device and the counters travel over gloo (RCCL refuses two ranks on one device)."""
    import torch
    two = torch.cuda.device_count() >= 2
    argv = ["--gpus", "2", "--groups", "64", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-points"]
    if not two:
        argv += ["--share-gpu", "--backend", "gloo"]
    d = _bench_line(argv)
    assert d["n_gpus"] == 2 and d["config"]["backend"] == ("nccl" if two else "gloo")
    # one GPU per rank: the counters go through the PRODUCT's reduction (lnsfaid_comm_init + lnsfaid_allreduce_counters, RCCL);
    # gloo only where two ranks share a device
    assert d["config"]["reduce"] == ("lnsfaid_allreduce_counters" if two else "torch.distributed gloo")
    assert len(d["config"]["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in d["config"]["per_rank_ms_per_step"])
    assert d["config"]["counters_TestFrame_ErrorFrame_ErrorBits_LT3"][0] == 2 * 64 * 32
    assert d["strong"]["counters_TestFrame_ErrorFrame_ErrorBits_LT3"][0] == 64 * 32
    assert d["value"] > 0 and d["roofline"]["launches"] >= 2


@pytest.mark.gpu
def test_bench_single_gpu_through_the_launcher():
    """N = 1 through the same spawn path the N > 1 runs take (RCCL world of one)."""
    d = _bench_line(["--gpus", "1", "--spawn", "--groups", "64", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-points"])
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl"
    assert d["config"]["reduce"] == "lnsfaid_allreduce_counters"  # RCCL communicator of one rank, through the C ABI
    assert d["config"]["counters_TestFrame_ErrorFrame_ErrorBits_LT3"][0] == 64 * 32
