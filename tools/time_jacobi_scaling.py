"""Jacobi stage time against the number of edges (1, 2, 4 workgroups per CU's worth) for both variants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_edge_trace_amd as amd
from bench import synth_image, README_KW
L = amd._lib
ctx = L.Context(0)
N = 500
img, truth = synth_image(N, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
for variant in [int(a) for a in sys.argv[1:]] or [0, 1]:
    L.set_option("jacobi_variant", variant)
    for E in [64, 256, 512, 768, 1024, 2048]:
        seeds = list(range(1, E + 1))
        tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
        tr._batch.iterate(seeds, 7)
        ms = tr._batch.profile_stage(122, 20)
        print("variant %d  %5d edges: %.3f ms" % (variant, E, ms), flush=True)
        del tr
