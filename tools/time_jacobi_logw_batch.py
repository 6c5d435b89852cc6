"""The rotation-log form of the LDS Jacobi (k_jacobi_seat<.., LOGW> + k_jacobi_wpass) on a BIG batch against the form that
keeps the eigenvectors in LDS: stage 122 (Jacobi) of the bench batch at its mid-trace state, and the whole loop.
GPET_JLOG_MAX_B=100000 python tools/time_jacobi_logw_batch.py 1024   (the log exists only for batches up to that size)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
for logw in (1, 0, 1, 0):
    L.set_option("jacobi_logw", logw)
    print("jacobi_logw=%d: stage 122 (Jacobi [+ eigenvector pass]) %.3f ms" % (logw, b.profile_stage(122, 20)), flush=True)
for logw in (1, 0, 1, 0):
    L.set_option("jacobi_logw", logw)
    tr.reset()
    ctx.sync()
    t0 = time.time()
    it = tr.run_loop()
    ctx.sync()
    print("jacobi_logw=%d: loop %.1f ms (%d..%d iterations)" % (logw, 1e3 * (time.time() - t0), min(it), max(it)), flush=True)
b.close()
