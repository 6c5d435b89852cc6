"""Sharding of independent edges over the GPUs of one node.

Edges share no state, so the partition is embarrassingly parallel: rank r traces a contiguous
block of edges; the only exchanges are one broadcast of the shared gradient image from rank 0
(RCCL over xGMI when the backend is "nccl") and one gather of the finished traces.  No
per-iteration collective exists.  The same code runs under "gloo" on CPU tensors (tests).
"""
from __future__ import annotations

import numpy as np


def edge_slice(n_edges, world, rank):
    """Contiguous block [lo, hi) of edge indices owned by ``rank`` (sizes differ by at most 1)."""
    base, rem = divmod(int(n_edges), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_array(arr, shape, dtype, dist, src=0, device="cpu"):
    """Broadcast a numpy array from ``src`` to every rank; returns it as numpy on all ranks."""
    import torch
    t = torch.empty(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), device=device)
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)))
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def gather_traces(local, n_edges, edge_len, dist, device="cpu"):
    """All ranks contribute their (n_local, edge_len, 2) int64 traces; every rank gets the
    (n_edges, edge_len, 2) array in global edge order."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    cap = max(edge_slice(n_edges, world, r)[1] - edge_slice(n_edges, world, r)[0] for r in range(world))
    buf = torch.zeros((cap, edge_len, 2), dtype=torch.int64, device=device)
    loc = np.asarray(local, dtype=np.int64).reshape(-1, edge_len, 2)
    if loc.shape[0]:
        buf[:loc.shape[0]].copy_(torch.from_numpy(loc))
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = []
    for r in range(world):
        lo, hi = edge_slice(n_edges, world, r)
        out.append(parts[r][:hi - lo].cpu().numpy())
    return np.concatenate(out, axis=0)


def trace_sharded(grad, grad_shape, inits, seeds, tracer, dist=None, device="cpu"):
    """Trace ``len(inits)`` independent edges of one shared gradient image across all ranks.

    ``grad`` is needed on rank 0 only.  ``tracer(grad, inits_block, seeds_block)`` returns the
    list of (edge_len, 2) traces of its block (GP_Edge_Tracing_Batch on a GPU; tests pass a CPU
    callable).  Returns the (n_edges, edge_len, 2) traces in global order on every rank."""
    n = len(inits)
    if dist is None or dist.get_world_size() == 1:
        return np.stack(tracer(grad, list(inits), list(seeds)))
    world, rank = dist.get_world_size(), dist.get_rank()
    grad = broadcast_array(grad, grad_shape, np.float32, dist, 0, device)
    lo, hi = edge_slice(n, world, rank)
    local = tracer(grad, list(inits[lo:hi]), list(seeds[lo:hi])) if hi > lo else []
    edge_len = int(abs(int(inits[0][-1][0]) - int(inits[0][0][0])) + 1)
    return gather_traces(local, n, edge_len, dist, device)
