"""Phase cycle counts of the seated Jacobi (library built with -DGPET_JAC_PROF, GPET_LIB_PATH pointing at it)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import gaussian_process_edge_trace_amd as amd
from bench import synth_image, README_KW
L = amd._lib
ctx = L.Context(0)
N = 500
img, truth = synth_image(N, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
for E in [int(a) for a in sys.argv[2:]] or [256, 1024]:
    seeds = list(range(1, E + 1))
    L.set_option("jacobi_variant", int(sys.argv[1]))
    tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
    tr._batch.iterate(seeds, 7)
    print("== %d edges, variant %s" % (E, sys.argv[1]), flush=True)
    ms = tr._batch.profile_stage(122, 1)
    ctx.sync()
    print("   %.3f ms" % ms, flush=True)
    del tr
