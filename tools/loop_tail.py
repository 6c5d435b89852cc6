"""How the device loop's time splits over iterations: histogram of iterations per trace over the bench batch and the
wall time of the loop when it is cut after k iterations (the tail iterations run with few active edges)."""
import sys
import time
import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
import gaussian_process_edge_trace_amd as pkg  # noqa: E402

L = pkg._lib
ctx = L.Context(0)
N = 500
img, truth = bench.synth_image(N, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
seeds = [1 + e for e in range(E)]
tr = pkg.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **bench.README_KW, _ctx=ctx)
tr.reset(); it = tr.run_loop()
print("iterations per trace: histogram", dict(zip(*np.unique(it, return_counts=True))), flush=True)
for chunk in (16, 32):
    ts = []
    for _ in range(5):
        tr.reset(); ctx.sync()
        t0 = time.time(); tr.run_loop(chunk=chunk); ts.append(time.time() - t0)
    print("run_loop chunk=%d: %.1f ms" % (chunk, 1e3 * min(ts)), flush=True)
for k in (4, 8, 12, 13, 14, 15, 16, 18, 20, 24, 32):
    ts = []
    for _ in range(3):
        tr.reset(); ctx.sync()
        t0 = time.time(); n_act = tr._batch.iterate(seeds, k); ts.append(time.time() - t0)
    print("first %2d iterations: %.1f ms, %d edges still active" % (k, 1e3 * min(ts), n_act), flush=True)
print("Jacobi sweeps of the factorisation of iteration k (edges 0..7):")
for k in range(1, 15):
    tr.reset()
    tr._batch.iterate(seeds, k)
    sc = tr._batch.all_scalars()
    print("  k=%2d n_obs=%3d sweeps" % (k, sc[0].n), [int(s.lml) for s in sc[:8]], "max over batch", max(int(s.lml) for s in sc), flush=True)
for look in (0, 1, 2, 4):
    pkg._lib.set_option("rng_lookahead", look)
    ts = []
    for _ in range(5):
        tr.reset(); ctx.sync()
        t0 = time.time(); tr.run_loop(); ts.append(time.time() - t0)
    print("rng_lookahead=%d: run_loop %.1f ms" % (look, 1e3 * min(ts)), flush=True)
pkg._lib.set_option("rng_lookahead", 0)
