"""Config 3's factor with the pivoted Cholesky in one workgroup (k_pchol) and over the GPU (k_pcx_step): rank, residual, time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_edge_trace_amd as amd
from bench import synth_image
L = amd._lib
ctx = L.Context(0)
N = 2048
img, truth = synth_image(N, 0)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
init = truth[[0, -1], :][:, [1, 0]]
rng = np.random.default_rng(0)
cols = np.sort(rng.choice(np.arange(1, N - 1), size=1498, replace=False))
obs = np.stack([cols, truth[cols, 0] + rng.integers(-2, 3, size=cols.size)], axis=1).astype(np.int64)
kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 300, 'length_scale': 80}, noise_y=1, N_samples=4000,
          score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, seed=1, fix_endpoints=True)
tr = amd.GP_Edge_Tracing(init, grad, obs=obs, **kw, _ctx=ctx)
b = tr._batch
b.set_obs(0, obs)
b.fit_predict(want_cov=True)
cov = b.read(L.BUF_COV)
for blocked in (0, 1, 2):
    L.set_option("pchol_multi", blocked)
    b.factor()
    ms = b.profile_stage(1, 5)
    s = b.scalars()
    A = b.read(L.BUF_FACTOR)
    G = b.read(L.BUF_G) if hasattr(L, "BUF_G") else None
    res = np.abs(A.T @ A - cov).max()
    print("pchol_multi %d: factor %.3f ms, rank %d, status %d, max |A^T A - cov| = %.3e (max |cov| %.3e, max diag %.3e)"
          % (blocked, ms, s.rank, s.status, res, np.abs(cov).max(), np.diag(cov).max()), flush=True)
    if G is not None:
        G = np.asarray(G)[:s.rank]
        print("   max |G^T G - cov| = %.3e" % np.abs(G.T @ G - cov).max())
    for st in (110, 111, 112, 113):
        print("   part %d: %.3f ms" % (st - 110, b.profile_stage(st, 5)))
