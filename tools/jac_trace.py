import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_edge_trace_amd as amd
from bench import synth_image, README_KW
L = amd._lib
ctx = L.Context(0)
E = 1024
img, truth = synth_image(500, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = list(range(1, E + 1))
tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
for it in range(1, 13):
    tr._batch.iterate(seeds, 1)
    ctx.sync()
    print("== after iteration", it, flush=True)
