"""Query sharding across ranks (one process per GPU) and the timing reduction bench.py uses.

Per-query output depends only on the index and on that query, so ranks take
contiguous slices of the query list balanced by token count; the only collective on the data
path is the one-time broadcast of the index.
"""
import numpy as np


def shard_bounds(qoff, ntok, world):
    """Contiguous query ranges [b[r], b[r+1]) with roughly equal token counts."""
    qoff = np.asarray(qoff, np.int64); nq = len(qoff)
    ends = np.concatenate((qoff[1:], [ntok]))
    targets = (np.arange(1, world) * ntok) // world
    cuts = np.searchsorted(ends, targets, side="left")
    cuts = np.minimum(np.maximum(cuts, 0), nq)
    b = np.concatenate(([0], cuts, [nq])).astype(np.int64)
    return np.maximum.accumulate(b)


def take_shard(qoff, qtok, rank, world):
    b = shard_bounds(qoff, len(qtok), world)
    q0, q1 = int(b[rank]), int(b[rank + 1])
    qoff = np.asarray(qoff, np.int64)
    t0 = int(qoff[q0]) if q0 < len(qoff) else len(qtok)
    t1 = int(qoff[q1]) if q1 < len(qoff) else len(qtok)
    return q0, (qoff[q0:q1] - t0).astype(np.int32), np.asarray(qtok[t0:t1], np.int32)


def max_over_ranks(seconds, dist=None):
    """Slowest rank's wall time (the job is done when the last shard is)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(value, dist=None):
    """Smallest value over the ranks (a decision every rank must take the same way: yes only if all say yes)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def sum_over_ranks(value, dist=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(value)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
