"""Time of one normal stream: sequential generator vs the chunked one. usage: python tools/time_normals.py [N] [S] [B]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from bench import synth_image  # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
L = amd._lib
ctx = L.Context(0)
img, truth = synth_image(N, 0)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
init = truth[[0, -1], :][:, [1, 0]]
kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1, N_samples=S,
          score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
for zc in (N, 96):
    batch = amd.GP_Edge_Tracing_Batch([init] * B, np.asarray(grad, dtype=np.float32), list(range(B)), **kw, _ctx=ctx, z_cols=zc)
    b = batch._batch
    res = {}
    for mode in (0, 1):
        L.set_option("rng_chunked", mode)
        b.normals(list(range(7, 7 + B)))
        ctx.sync()
        t0 = time.time()
        for _ in range(5):
            b.normals(list(range(7, 7 + B)))
        ctx.sync()
        res[mode] = (1e3 * (time.time() - t0) / 5, b.read(L.BUF_NORMALS))
    L.set_option("rng_chunked", -1)
    print("N=%d S=%d B=%d z_cols=%d: sequential %.3f ms, chunked %.3f ms, identical %s" %
          (N, S, B, zc, res[0][0], res[1][0], np.array_equal(res[0][1], res[1][1])), flush=True)
    b.close()
