"""Warm start of the structured path's eigen-decomposition (option jacobi_warm): whole traces of a batch of README edges
with and without it -- wall time of the batch alone (loop + converged fits), Jacobi sweeps of the last iteration, and
whether the traces are the same.  (A single kernel cannot be timed for this: repeated launches on a fixed state start
from the eigenvectors of the very matrix they factor.)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from bench import synth_image, README_KW  # noqa: E402
L = amd._lib
ctx = L.Context(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
img, truth = synth_image(500, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = [1 + 997 * k for k in range(E)]
ref = None
for warm in (0, 1, 0, 1):
    old = L.set_option("jacobi_warm", warm)
    best = 1e9
    for rep in range(3):
        tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
        t0 = time.time()
        traces = tr()
        dt = time.time() - t0
        best = min(best, dt)
        loop_ms = 1e3 * tr.timings.get("loop_s", 0.0)
        iters = list(tr.timings["iters"])
        tr._batch.close()
    L.set_option("jacobi_warm", old)
    if ref is None:
        ref = traces
    same = sum(int(np.array_equal(a, b)) for a, b in zip(traces, ref))
    print("jacobi_warm %d: %.1f ms per batch of %d traces (best of 3; loop %.1f ms), iterations %.2f mean; traces equal to the cold start's: %d of %d"
          % (warm, 1e3 * best, E, loop_ms, float(np.mean(iters)), same, E), flush=True)
