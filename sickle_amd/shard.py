"""Multi-GPU sharding of the scan: reads are independent, so each rank (one process per GPU)
takes a contiguous range of the batch and nothing is exchanged on the data path.  The only
cross-rank step is the sum of the kept/discarded counters (SE: 2, PE: 6 integers) and the
max of the elapsed times, done once after the timed region."""


def shard_range(n_total, rank, world):
    """Contiguous [begin, end) of rank's reads; sizes differ by at most one."""
    base, extra = divmod(int(n_total), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def reduce_counters(dist, counters, elapsed, device=None):
    """Sum the integer counters and take the max of elapsed over all ranks.
    `dist` is torch.distributed (initialised) or None for a single process."""
    import torch
    c = torch.tensor([int(x) for x in counters], dtype=torch.int64, device=device)
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(x) for x in c.tolist()], float(t.item())
