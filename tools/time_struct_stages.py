import sys, os
sys.path.insert(0, os.getcwd())
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
b = tr._batch
b.iterate(seeds, 7)
print("shared basis off" if os.environ.get("GPET_SHARED_BASIS") == "0" else "shared basis on", " ".join("%d: %.3f ms" % (k, b.profile_stage(k, 20)) for k in (120, 121, 122, 123)), flush=True)
