"""The fused sample + score kernel (stage 131) and the kept rows (132) of the bench batch at its mid-trace state, a few
launches each: the program rocprofv3 / tools/pmc_kernel.sh runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_edge_trace_amd as amd
from bench import synth_image, README_KW
L = amd._lib
ctx = L.Context(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
img, truth = synth_image(500, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = list(range(1, E + 1))
tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
b = tr._batch
b.iterate(seeds, 7)
for k in (131, 132, 130, 140):
    print(k, "%.3f ms" % b.profile_stage(k, 3), flush=True)
b.close()
