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


def broadcast_tensor(arr, shape, dtype, dist, src=0, device="cpu"):
    """Broadcast a numpy array from ``src`` to every rank; returns the torch tensor the collective filled (on
    ``device``: with backend "nccl" that is this rank's GPU, and the tracer consumes it in place through
    ``tensor.data_ptr()`` -- gpet_batch_create2 / GPET_GRAD_ON_DEVICE -- with no trip through host memory)."""
    import torch
    t = torch.empty(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), device=device)
    if dist.get_rank() == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)))
    dist.broadcast(t, src=src)
    if t.is_cuda:
        # the collective (and rank src's host-to-device copy) only order torch's current stream; the tracer reads the
        # tensor on the library's own HIP stream through data_ptr(): wait here, once, before handing it out
        torch.cuda.current_stream(t.device).synchronize()
    return t


def broadcast_array(arr, shape, dtype, dist, src=0, device="cpu"):
    """Broadcast a numpy array from ``src`` to every rank; returns it as numpy on all ranks."""
    return broadcast_tensor(arr, shape, dtype, dist, src, device).cpu().numpy()


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
    g = broadcast_tensor(grad, grad_shape, np.float32, dist, 0, device)
    # on a GPU the tracer gets the device tensor itself (it passes data_ptr() to the library); on CPU, numpy
    grad = g if str(g.device).startswith("cuda") else g.numpy()
    lo, hi = edge_slice(n, world, rank)
    local = tracer(grad, list(inits[lo:hi]), list(seeds[lo:hi])) if hi > lo else []
    edge_len = int(abs(int(inits[0][-1][0]) - int(inits[0][0][0])) + 1)
    return gather_traces(local, n, edge_len, dist, device)


def trace_sharded_cabi(grad, grad_shape, inits, seeds, comm, **ctor_kwargs):
    """``trace_sharded`` with no torch in the process: the C ABI's own RCCL call sites (``_lib.Comm``: gpet_comm_create,
    gpet_bcast_grad, gpet_gather_traces; include/gpet_hip.h "collectives") -- what a non-Python host of the reference would
    call.  ``grad`` is needed on rank 0 only; the broadcast image is consumed where RCCL put it (device pointer ->
    gpet_batch_create2 / GPET_GRAD_ON_DEVICE).  Returns the (n_edges, edge_len, 2) traces in global order on every rank."""
    from .gpet import GP_Edge_Tracing_Batch
    n = len(inits)
    ptr = comm.bcast_grad(grad, grad_shape, root=0)
    lo, hi = comm.block(n)
    local = []
    if hi > lo:
        local = GP_Edge_Tracing_Batch(list(inits[lo:hi]), None, list(seeds[lo:hi]), grad_device_ptrs=[ptr],
                                      grad_shape=tuple(grad_shape), _ctx=comm.ctx, **ctor_kwargs)()
    edge_len = int(abs(int(inits[0][-1][0]) - int(inits[0][0][0])) + 1)
    return comm.gather_traces(local, n, edge_len)


def sequence_partition(n_frames, n_chains, world, rank):
    """Frames of an image sequence owned by ``rank``: the sequence is cut into ``n_chains`` chains of consecutive
    frames (``sequence.chain_slices``; the first frame of a chain starts cold, later ones warm-start from the
    previous trace), and whole chains are dealt to the ranks in contiguous blocks like independent edges.
    Returns (first_frame, last_frame_exclusive, number_of_local_chains)."""
    from .sequence import chain_slices
    chains = chain_slices(n_frames, n_chains)
    lo, hi = edge_slice(len(chains), world, rank)
    if hi <= lo:
        return 0, 0, 0
    return chains[lo][0], chains[hi - 1][1], hi - lo


def trace_sequence_sharded(frames, frame_shape, n_frames, init, n_chains, tracer, dist=None, device="cpu"):
    """Trace one edge through ``n_frames`` gradient images in ``n_chains`` chains spread over all ranks
    (BASELINE config 5: 64 frames, 8 chains, 8 GPUs).  ``frames`` ((T, M, N) float32) is needed on rank 0 only and is
    broadcast once; ``tracer(frames_block, first_frame, n_local_chains)`` returns the (edge_len, 2) traces of a
    contiguous block of frames that consists of whole chains.  Every rank gets the (T, edge_len, 2) traces."""
    if dist is None or dist.get_world_size() == 1:
        return np.stack(tracer(frames, 0, min(int(n_chains), int(n_frames))))
    world, rank = dist.get_world_size(), dist.get_rank()
    t = broadcast_tensor(frames, (n_frames,) + tuple(frame_shape), np.float32, dist, 0, device)
    f0, f1, nc = sequence_partition(n_frames, n_chains, world, rank)
    block = t[f0:f1] if str(t.device).startswith("cuda") else t[f0:f1].numpy()
    local = tracer(block, f0, nc) if nc else []
    # gather: frame blocks are contiguous and in rank order, but of different lengths -> pad to the longest
    import torch
    edge_len = int(abs(int(init[-1][0]) - int(init[0][0])) + 1)
    sizes = [sequence_partition(n_frames, n_chains, world, r) for r in range(world)]
    cap = max(b - a for a, b, _ in sizes)
    buf = torch.zeros((cap, edge_len, 2), dtype=torch.int64, device=device)
    loc = np.asarray(local, dtype=np.int64).reshape(-1, edge_len, 2)
    if loc.shape[0]:
        buf[:loc.shape[0]].copy_(torch.from_numpy(loc))
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return np.concatenate([parts[r][:sizes[r][1] - sizes[r][0]].cpu().numpy() for r in range(world)], axis=0)
