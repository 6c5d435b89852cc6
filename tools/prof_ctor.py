"""The constructor of a 1 024-edge batch alone (what `secondary.ctor_included` adds to a step): host wall time per call, with cProfile's
view of the Python side.  usage: python tools/prof_ctor.py [edges] [reps]   (under rocprofv3 --hip-trace --kernel-trace --stats for the C side)"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_edge_trace_amd as pkg  # noqa: E402
from bench import synth_image, README_KW  # noqa: E402
L = pkg._lib
ctx = L.Context(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
img, truth = synth_image(500, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = list(range(1, E + 1))
pkg.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)._batch.close()
pr = cProfile.Profile()
ts = []
for _ in range(reps):
    ctx.sync()
    t0 = time.time()
    pr.enable()
    b = pkg.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
    ctx.sync()
    pr.disable()
    ts.append(time.time() - t0)
    t0 = time.time()
    b._batch.close()
    ctx.sync()
    tc = time.time() - t0
print("constructor of %d edges: %s ms; close %.1f ms" % (E, ", ".join("%.1f" % (1e3 * t) for t in ts), 1e3 * tc))
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
