"""Multi-GPU plumbing of the decode path: groups shard trivially, counters are summed.

The reference sums 4 counters per worker thread on the main thread after pthread_join
(reference main.cpp:174-182).  One process per GPU does the same with one all-reduce of a 4 x int64 tensor
over torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests).  No other traffic
crosses GPUs: a group of 32 codewords is never split (its early-stop rule couples its lanes).
"""


def shard_groups(n_groups, rank, world):
    """Contiguous range [first, last) of whole groups owned by `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    first = n_groups * rank // world
    last = n_groups * (rank + 1) // world
    return first, last


def allreduce_counters(counters, dist=None, device=None):
    """Sum [TestFrame, ErrorFrame, ErrorBits, LT3ErrBitFrame] over all ranks; identity without a process group."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(c) for c in counters]
    import torch
    t = torch.tensor([int(c) for c in counters], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]
