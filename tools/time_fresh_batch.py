import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import gaussian_process_edge_trace_amd as pkg
from bench import synth_image, README_KW
L = pkg._lib
ctx = L.Context(0)
N = 500
img, truth = synth_image(N, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
E = 1024
seeds = list(range(1, E + 1))
for rep in range(4):
    t1 = time.time()
    fresh = pkg.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
    t2 = time.time()
    fresh.reset(); it = fresh.run_loop(); ctx.sync()
    t3 = time.time()
    fresh.finish(it); ctx.sync()
    t4 = time.time()
    print("ctor %.3f loop %.3f fit %.3f" % (t2 - t1, t3 - t2, t4 - t3), flush=True)
    fresh._batch.close()
