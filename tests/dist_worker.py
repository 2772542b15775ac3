"""Worker of tests/test_dist_gloo.py: one rank of a world_size-N gloo job on CPU.  The decoder stand-in is
the oracle (this is a CPU test of the sharding + counter reduction, not of the kernels)."""
import importlib.util
import json
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_abi as oa  # noqa: E402

spec = importlib.util.spec_from_file_location("lnsfaid_dist", os.path.join(oa.PKG_DIR, "dist.py"))
lnsfaid_dist = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lnsfaid_dist)

if __name__ == "__main__":
    n_groups, eb_n0, out_path = int(sys.argv[1]), float(sys.argv[2]), sys.argv[3]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    code = oa.pyabi.Code50GPON()
    cfg = oa.pyabi.default_cfg(2, 10)
    fix = oa.ReferenceChannel(code, 101, 13.0).groups(eb_n0, n_groups)  # every rank regenerates the same batch
    first, last = lnsfaid_dist.shard_groups(n_groups, rank, world)
    per = 32 * code.N
    local = [0, 0, 0, 0]
    if last > first:
        dec, _ = oa.decode_mt(code, cfg, fix[first * per:last * per], last - first, threads=2)
        local = oa.Oracle(code, cfg).count_errors(dec, None, last - first)
    total = lnsfaid_dist.allreduce_counters(local, dist)
    dist.barrier()
    if rank == 0:
        json.dump({"total": total, "world": world}, open(out_path, "w"))
    dist.destroy_process_group()
