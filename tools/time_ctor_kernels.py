import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import gaussian_process_edge_trace_amd as amd
from bench import synth_image
ctx = amd._lib.Context(0)
for N in (2048, 1024, 500):
    img, truth = synth_image(N, 0)
    k = amd.gpet_utils.kernel_builder((11, 5))
    amd.gpet_utils.comp_grad_img(img, k, ctx=ctx)
    ctx.sync(); t0 = time.time()
    for _ in range(5): grad = amd.gpet_utils.comp_grad_img(img, k, ctx=ctx)
    ctx.sync(); print("N=%d comp_grad_img %.2f ms (host to host)" % (N, 1e3 * (time.time() - t0) / 5))
    init = truth[[0, -1], :][:, [1, 0]]
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, noise_y=1, N_samples=256, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, seed=1, fix_endpoints=True)
    tr = amd.GP_Edge_Tracing(init, grad, **kw, _ctx=ctx)
    b = tr._batch
    g32 = [np.asarray(grad).astype(np.float32)]
    b.set_images(g32); ctx.sync(); t0 = time.time()
    for _ in range(5): b.set_images(g32)
    ctx.sync(); print("N=%d set_images (upload + gradient KDE + reset) %.2f ms" % (N, 1e3 * (time.time() - t0) / 5))
