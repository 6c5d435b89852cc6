"""For rocprofv3 --kernel-trace --stats: cold eigen-decompositions (stage 122, warm start off) of the bench edge's mid-trace
matrix with k_jacobi_seat (variant 0) and k_jacobi_ahead (1).  usage: python tools/prof_jacobi_trace.py [edges] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from bench import synth_image, README_KW  # noqa: E402
L = amd._lib
ctx = L.Context(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
img, truth = synth_image(500, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = [1 + 997 * k for k in range(E)]
tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
b = tr._batch
b.iterate(seeds, 7)
L.set_option("jacobi_warm", 0)
for variant in (0, 1):
    L.set_option("jacobi_variant", variant)
    ms = b.profile_stage(122, reps)
    print("variant %d, %d edge(s): eigen stage %.4f ms, sweeps %d" % (variant, E, ms, int(b.scalars(0).lml)), flush=True)
