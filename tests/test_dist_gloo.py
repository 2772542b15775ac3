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
