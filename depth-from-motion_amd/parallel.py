"""Data-parallel glue of the path: the batch shards by sample,
one process per GPU, forward has NO collective; training adds one bucketed
gradient all-reduce per step (what MMDistributedDataParallel does in the
reference, apis/train.py:222-230) over RCCL (`backend='nccl'` on ROCm).

Only torch.distributed is used; the same code runs on gloo for the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(num_samples, rank, world_size):
    """Contiguous slice of a global batch owned by `rank` (remainder to the
    first ranks), e.g. 64 samples on 8 ranks -> 8 each."""
    base, extra = divmod(num_samples, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_over_ranks(value, device=None):
    """Slowest-rank time (bench.py's wall clock)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_gradients(parameters, bucket_bytes=64 << 20, average=True):
    """Sum (then average) .grad over ranks in flat buckets.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce
    is per-link bound: few large buckets amortise the ring latency better than
    DDP's 25 MB default; the KITTI student (~40 MB of grads) goes in one bucket.
    Returns the number of buckets reduced."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads or not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size()
    buckets, cur, cur_bytes = [], [], 0
    for g in grads:
        nbytes = g.numel() * g.element_size()
        if cur and (cur_bytes + nbytes > bucket_bytes or g.dtype != cur[0].dtype):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(g)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    for bucket in buckets:
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= world
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    return len(buckets)
