"""Multi-GPU plumbing: one process per GPU, channels sharded, no data-path collective.

Every channel is an independent stream (nothing in ProcessIQData couples channels), so ranks
never exchange samples.  The only collective is the one-shot broadcast of the coefficient blob
after a filter change (rank 0 designs, everyone installs): RCCL over xGMI on GPUs ("nccl"
backend), gloo in the CPU tests.  SURVEY 8e.
"""
import numpy as np


def shard_channels(n_total, rank, world):
    """contiguous channel range [lo, hi) owned by `rank` (sizes differ by at most one)"""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_coeffs(blob, src=0, device=None, group=None):
    """Broadcast a coefficient blob (numpy uint8, from RxChain.coeffs()/design_coeffs) from rank
    `src` to every rank; returns the received blob as numpy uint8.  Call on all ranks with a
    buffer of the right size (non-src contents are overwritten)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.uint8).copy())
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src, group=group)
    return t.cpu().numpy()


def max_over_ranks(x, device=None, group=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
