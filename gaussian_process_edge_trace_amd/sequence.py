"""Image-sequence tracing (BASELINE config 5): one edge followed through the frames of a sequence.

The reference traces one image per ``GP_Edge_Tracing`` object; a sequence is chained by the user through the
constructor's ``obs`` argument (gp_edge_tracing/gpet.py:57-61, 100, 820): pixels of the previous frame's trace are the
warm-start observations of the next.  With ``algo_thresh`` or more of them the while-loop would not run at all
(gpet.py:829), so the warm start takes every ``warm_every``-th pixel of the previous trace and must stay below it.

A chain is serial by construction (frame t needs trace t-1).  Parallelism comes from independent CHAINS: a sequence
of T frames is cut into C chains of consecutive frames, the first frame of every chain starting cold;
step s of all chains is one batch of C edges on the GPU (``GP_Edge_Tracing_Batch``, one image per edge), and chains
spread over the GPUs of a node like independent edges do (``sharding.trace_sequence_sharded``).
"""
from __future__ import annotations

import numpy as np

from .gpet import GP_Edge_Tracing_Batch, resolve_params


def chain_slices(n_frames, n_chains):
    """[lo, hi) frame ranges of ``n_chains`` chains of consecutive frames (lengths differ by at most one)."""
    n_chains = max(1, min(int(n_chains), int(n_frames)))
    base, rem = divmod(int(n_frames), n_chains)
    out, lo = [], 0
    for c in range(n_chains):
        hi = lo + base + (1 if c < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def warm_start_obs(edge_trace, x_st, x_en, warm_every, algo_thresh, M=None):
    """Observations (xy int64) for the next frame: every ``warm_every``-th pixel of ``edge_trace`` ((N, 2) yx, as
    ``GP_Edge_Tracing.__call__`` returns it) strictly inside the end points.  Fewer than ``algo_thresh`` of them, or
    the next frame's loop would be skipped (gpet.py:829): a too dense choice is thinned by doubling the stride."""
    et = np.asarray(edge_trace)
    step = max(1, int(warm_every))
    while True:
        sel = et[step:-1:step] if step < et.shape[0] else et[:0]
        sel = sel[(sel[:, 1] > x_st) & (sel[:, 1] < x_en)]
        if M is not None:  # (a rounded posterior mean may leave the image; such a pixel cannot be an observation)
            sel = sel[(sel[:, 0] >= 0) & (sel[:, 0] <= M - 1)]
        if sel.shape[0] < algo_thresh or sel.shape[0] == 0:
            return sel[:, [1, 0]].astype(np.int64)
        step *= 2


class SequenceTracer(object):
    """Traces ``init`` through ``frames`` (T gradient images of one shape) in ``n_chains`` chains on one GPU.

    ``frames``: sequence of (M, N) gradient images; ``seeds``: one seed per frame (default: ``seed`` for all, like a
    user re-creating ``GP_Edge_Tracing(..., seed=seed)`` per frame).  Remaining keyword arguments are the reference
    constructor's (gpet.py:22-35).  ``__call__`` returns the list of T results in frame order, each what
    ``GP_Edge_Tracing.__call__`` returns for that frame (trace, or (trace, credible interval) with ``return_std``)."""

    def __init__(self, frames, init, n_chains=1, warm_every=None, seed=42, seeds=None, *, device=0, _ctx=None, **kw):
        self.frames = frames
        self.T = len(frames)
        self.init = np.asarray(init)
        self.kw = dict(kw)
        self.kw.pop("obs", None)
        self.chains = chain_slices(self.T, n_chains)
        self.seeds = [int(seed)] * self.T if seeds is None else [int(v) for v in seeds]
        p = resolve_params(self.init, np.asarray(frames[0]).shape, **{k: v for k, v in self.kw.items()
                                                                      if k in ("kernel_options", "noise_y", "N_samples", "score_thresh",
                                                                               "delta_x", "keep_ratio", "pixel_thresh", "return_std",
                                                                               "fix_endpoints")})
        self._p = p
        self.warm_every = int(warm_every) if warm_every else 2 * p["delta_x"]
        self.device, self._ctx = device, _ctx
        self.iterations = [0] * self.T
        self._tracer = None

    def _frames_of_step(self, s):
        return [lo + s for lo, hi in self.chains if lo + s < hi]

    def __call__(self, max_iter=1000):
        results = [None] * self.T
        prev = {}  # chain index -> previous edge trace
        n_steps = max(hi - lo for lo, hi in self.chains)
        C = len(self.chains)
        for s in range(n_steps):
            active = [(c, lo + s) for c, (lo, hi) in enumerate(self.chains) if lo + s < hi]
            obs = []
            for c, f in active:
                obs.append(np.zeros((0, 2), dtype=np.int64) if s == 0 else
                           warm_start_obs(prev[c], self._p["x_st"], self._p["x_en"], self.warm_every, self._p["algo_thresh"],
                                          self._p["M"]))
            imgs = [np.asarray(self.frames[f]) for _, f in active]
            seeds = [self.seeds[f] for _, f in active]
            if self._tracer is None or len(active) != self._tracer.B:
                # (first step, or the shorter chains have run out: a smaller batch from here on; the old batch's arena,
                # streams and events are released now, not whenever the garbage collector gets to them)
                if self._tracer is not None:
                    self._tracer._batch.close()
                self._tracer = GP_Edge_Tracing_Batch([self.init] * len(active), imgs, seeds, obs=obs, device=self.device,
                                                     _ctx=self._ctx, **self.kw)
                if self._ctx is None:
                    self._ctx = self._tracer._ctx
            else:
                self._tracer.set_frame(imgs, obs, seeds)
            out = self._tracer(max_iter)
            iters = self._tracer.timings["iters"]
            for k, (c, f) in enumerate(active):
                results[f] = out[k]
                self.iterations[f] = iters[k]
                prev[c] = out[k][0] if self._tracer.return_std else out[k]
        return results


def trace_sequence(frames, init, n_chains=1, warm_every=None, **kw):
    """Convenience wrapper: ``SequenceTracer(frames, init, n_chains, warm_every, **kw)()``."""
    return SequenceTracer(frames, init, n_chains, warm_every, **kw)()
