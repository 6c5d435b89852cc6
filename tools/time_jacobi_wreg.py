"""k_jacobi_seat with 0 / 28 / 42 / 56 rows of the eigenvector matrix in registers (option jacobi_wreg): device time of the
kernel on the bench batch at its mid-trace state, and the factor against the all-LDS form (must be bit-identical)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import gaussian_process_edge_trace_amd as amd  # noqa: E402
import bench  # noqa: E402
from bench import synth_image, README_KW  # noqa: E402
L = amd._lib
ctx = L.Context(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
img, truth = synth_image(500, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = list(range(1, E + 1))
tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
tr._batch.iterate(seeds, 7)
ids = dict(bench.KERNEL_IDS_STRUCT)
ids.update(bench.KERNEL_IDS_COMMON)
kid = [k for k, v in ids.items() if v == "k_jacobi_seat"][0]
kid_rows = [k for k, v in ids.items() if v == "k_struct_rows"][0]
ref = None
for rr in (0, 4, 6, 8):
    old = L.set_option("jacobi_wreg", rr)
    ms = tr._batch.profile_stage(kid, 10)
    tr._batch.profile_stage(kid_rows, 1)
    fac = [np.array(tr._batch.read(L.BUF_FACTOR, e)) for e in (0, E // 2, E - 1)]
    L.set_option("jacobi_wreg", old)
    if ref is None:
        ref = fac
    same = all(np.array_equal(a, b) for a, b in zip(fac, ref))
    print("jacobi_wreg %d: k_jacobi_seat %.3f ms per launch of %d edges; factor rows identical to the LDS form: %s" % (rr, ms, E, same), flush=True)
