"""How tall is the band of image rows that the samples of a 32-column group can touch (mean +- 6.5 sigma), iteration by
iteration of the bench trace?  (Sizing of a band-limited image slab for scoring inside the sample GEMM.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_edge_trace_amd as amd
from bench import synth_image, README_KW
L = amd._lib
ctx = L.Context(0)
E = 8
img, truth = synth_image(500, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = list(range(1, E + 1))
tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
b = tr._batch
for it in range(1, 15):
    n_act = b.iterate(seeds, 1)
    rows = []
    for e in range(E):
        sc = b.scalars(e)
        A = b.read(L.BUF_FACTOR, e)
        mu = b.read(L.BUF_MEAN, e)
        Y = b.read(L.BUF_SAMPLES, e)
        sd = np.sqrt((A ** 2).sum(axis=0))
        ys = Y.std(axis=0)
        lo = np.floor(Y.mean(axis=0) - 6.5 * ys) - 1
        hi = np.ceil(Y.mean(axis=0) + 6.5 * ys) + 2
        h = []
        for g in range(0, 500, 32):
            h.append(max(hi[g:g + 33]) - min(lo[g:g + 33]) + 1)
        rows.append((np.median(h), max(h), float(np.median(ys)), float(ys.max()), float(np.median(sd)), sc.n_obs))
    r = np.array(rows)
    print("iteration %2d: band height per 32-column group median %5.0f max %5.0f | sample std median %.2f max %.2f px (factor col-norm median %.3f) n_obs %d"
          % (it, np.median(r[:, 0]), r[:, 1].max(), np.median(r[:, 2]), r[:, 3].max(), np.median(r[:, 4]), int(r[0, 5])), flush=True)
    if n_act == 0:
        break
