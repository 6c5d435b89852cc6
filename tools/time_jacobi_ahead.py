"""k_jacobi_ahead (jacobi_variant 1: parameters one round ahead, one barrier per round) against k_jacobi_seat (0) on the bench
edge at its mid-trace state: time of the eigen-decomposition stage (pre-rotation + Jacobi + rotation-log pass where the batch
has one), eigenvalues, factor rows and whole traces.  usage: python tools/time_jacobi_ahead.py [edges ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from bench import synth_image, README_KW  # noqa: E402

L = amd._lib
ctx = L.Context(0)
N = 500
img, truth = synth_image(N, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
for E in [int(v) for v in sys.argv[1:]] or [1, 32, 1024]:
    seeds = [1 + 997 * k for k in range(E)]
    res = {}
    for variant in (0, 1):
        old = L.set_option("jacobi_variant", variant)
        try:
            tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
            b = tr._batch
            b.iterate(seeds, 7)
            sc = b.scalars(0)
            # (a repeated launch on a fixed state would start warm from the eigenvectors of the very matrix it factors: cold here)
            oldw = L.set_option("jacobi_warm", 0)
            b.profile_stage(122, 2)
            ms = b.profile_stage(122, 20)
            L.set_option("jacobi_warm", oldw)
            sweeps = [int(s.lml) for s in b.all_scalars()]
            b.profile_stage(123, 1)
            fac = np.array(b.read(L.BUF_FACTOR, 0))[:sc.rank]
            tr.reset()
            t0 = time.time()
            out = tr()
            dt = time.time() - t0
            res[variant] = (ms, fac, out, list(tr.timings["iters"]))
            print("E=%d variant %d: eigen stage %.4f ms (rank %d, sweeps %d..%d); whole traces %.3f s (loop %.3f s)"
                  % (E, variant, ms, sc.rank, min(sweeps), max(sweeps), dt, tr.timings["loop_s"]), flush=True)
            b.close()
        finally:
            L.set_option("jacobi_variant", old)
    f1, f2 = res[0][1], res[1][1]
    print("E=%d: factor rows max |diff| %.3e (scale %.3e); traces identical: %d of %d; iterations identical: %s"
          % (E, np.abs(f1 - f2).max(), np.abs(f1).max(), sum(np.array_equal(a, c) for a, c in zip(res[0][2], res[1][2])), E, res[0][3] == res[1][3]), flush=True)
