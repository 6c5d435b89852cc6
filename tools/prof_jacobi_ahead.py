"""Cycle stamps of k_jacobi_ahead's rounds (library built with tools/build_instrumented.sh -DGPET_JAC_PROF): a cold
eigen-decomposition of the bench edge's mid-trace matrix.  usage: GPET_LIB_PATH=... python tools/prof_jacobi_ahead.py [edges]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from bench import synth_image, README_KW  # noqa: E402
L = amd._lib
ctx = L.Context(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1
img, truth = synth_image(500, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = [1 + 997 * k for k in range(E)]
tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
b = tr._batch
L.set_option("jacobi_variant", 0)
b.iterate(seeds, 7)
L.set_option("jacobi_warm", 0)
for variant in (0, 1):
    L.set_option("jacobi_variant", variant)
    ms = b.profile_stage(122, 1)
    print("variant %d, %d edge(s): eigen stage %.4f ms, sweeps %d" % (variant, E, ms, int(b.scalars(0).lml)), flush=True)
