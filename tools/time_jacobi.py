"""Time the LDS Jacobi variants (gpet_set_option "jacobi_variant") on the bench batch at its mid-trace state and
compare what they produce: eigenvalues, factor rows after the rows stage, and whole traces."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_edge_trace_amd as amd
from bench import synth_image, README_KW

L = amd._lib
ctx = L.Context(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = 500
img, truth = synth_image(N, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = list(range(1, E + 1))
res = {}
for variant in [0, 1]:
    L.set_option("jacobi_variant", variant)
    tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
    tr._batch.iterate(seeds, 7)
    sc = tr._batch.scalars(0)
    ms = tr._batch.profile_stage(122, 20)
    tr._batch.profile_stage(123, 1)
    fac = tr._batch.read(L.BUF_FACTOR, 0)[:sc.rank]
    tr.reset()
    t0 = time.time(); out = tr(); dt = time.time() - t0
    res[variant] = (ms, fac, out)
    print("variant %d: jacobi %.3f ms  (rank %d, sweeps %d)  full trace of %d edges %.3f s" % (variant, ms, sc.rank, int(sc.lml), E, dt), flush=True)
    del tr
for v in [1]:
    f0, f1 = res[0][1], res[v][1]
    print("variant %d vs 0: factor rows max |diff| %.3e (scale %.3e)" % (v, np.abs(f0 - f1).max(), np.abs(f0).max()))
    same = sum(np.array_equal(a, b) for a, b in zip(res[0][2], res[v][2]))
    print("variant %d vs 0: %d of %d traces identical" % (v, same, E))
