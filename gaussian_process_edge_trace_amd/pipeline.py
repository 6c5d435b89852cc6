"""Several batches of edges in flight on one GPU.

One ``GP_Edge_Tracing_Batch`` keeps the GPU busy but not full: the last iterations of its device loop run with few edges
left, every kernel has a tail of partly filled CUs, and the converged fits end in rounds of small launches.  With three
or more batch objects in flight -- each on its own HIP stream (its own ``_lib.Context``), each driven by its own host
thread through reset -> device loop -> converged fits -- the GPU always has other work to fill itself from: 7.3 k
instead of 6.6-6.9 k edge-traces/s on 500 x 500 edges (bench.py schedules its steps this way).
The host threads only enqueue launches and wait (ctypes releases the GIL inside the library).
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor


def step_owner(n_steps, n_tracers):
    """Which batch object runs which step: step k on object k mod W, W = min(objects, steps)."""
    w = max(1, min(int(n_tracers), int(n_steps)))
    return [[k for k in range(wi, int(n_steps), w)] for wi in range(w)]


def run_in_flight(tracers, n_steps, prepare=None, executor=None):
    """``n_steps`` whole traces over the batch objects in ``tracers``, len(tracers) of them in flight.

    Step k runs on ``tracers[k % W]`` in that object's own thread: ``prepare(tracer, k)`` (optional: new images through
    ``set_frame``, new seeds, ...; the default resets the object to its constructor state), then ``tracer()`` --
    device loop and converged fits.  Returns the results of the steps in step order.  An object is only ever touched by
    its own thread; the objects must have been built on different ``_lib.Context`` objects (= HIP streams) of one device.
    """
    tracers = list(tracers)
    plan = step_owner(n_steps, len(tracers))
    results = [None] * int(n_steps)

    def worker(w):
        tr = tracers[w]
        for k in plan[w]:
            if prepare is not None:
                prepare(tr, k)
            else:
                tr.reset()
            results[k] = tr()
        return w

    own = executor is None
    ex = executor if executor is not None else ThreadPoolExecutor(max_workers=len(plan))
    try:
        list(ex.map(worker, range(len(plan))))  # (re-raises a worker's exception here)
    finally:
        if own:
            ex.shutdown()
    return results
