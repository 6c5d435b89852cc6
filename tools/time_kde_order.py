import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
tr._batch.iterate(seeds, 7)
p = tr._batch.profile_stage
for label, seq in [("151 x20", [(151, 20)]), ("150 x20", [(150, 20)]), ("5 x20", [(5, 20)]), ("151 x1", [(151, 1)]), ("150 x1", [(150, 1)]),
                   ("151 x1", [(151, 1)]), ("5 x1", [(5, 1)]), ("5 x1", [(5, 1)]), ("4 x1 then 151 x1", [(4, 1), (151, 1)]), ("151 x20", [(151, 20)])]:
    print(label, ["%.3f" % p(s, r) for s, r in seq], flush=True)
