"""Device loop of one 1024-edge batch object alone (no pipelining), with numpy's generator and with the Philox mode."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_edge_trace_amd as pkg
from bench import synth_image, README_KW
L = pkg._lib
ctx = L.Context(0)
N = 500
img, truth = synth_image(N, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
E = 1024
seeds = list(range(1, E + 1))
for rng in ("mt19937", "philox", "mt19937"):
    tr = pkg.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx, rng=rng)
    for rep in range(3):
        tr.reset(); ctx.sync()
        t0 = time.time(); it = tr.run_loop(); ctx.sync(); t1 = time.time()
    print("%s: loop %.1f ms (%d iterations max)" % (rng, 1e3 * (t1 - t0), max(it) if hasattr(it, '__iter__') else it), flush=True)
    tr._batch.close()
