"""Warm-started any-rank factor (the case of every iteration but a trace's first) against the stopping tolerance: sweeps, time,
orthogonality of the rows and distance of the samples from LAPACK's.  One process per tolerance (options are per batch).
usage: GPET_OJ_TOL_EXP=<x> python tools/tolsweep_warm.py [N=1024]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_edge_trace_amd as amd
from oracle import gpet_oracle as orc
L = amd._lib; ctx = L.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
img, truth = orc.synth_sinusoid_image(N, 5)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
init = truth[[0, -1], :][:, [1, 0]]
kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1, N_samples=300,
          score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
Z = orc.legacy_standard_normal(11, 64 * N).reshape(64, N)
bt = amd.GP_Edge_Tracing_Batch([init], grad, [3], **kw, _ctx=ctx)
b = bt._batch
for it in range(1, 7):
    b.iterate([3], 1)
    s = b.scalars(0)
    if s.done:
        break
    A = np.array(b.read(L.BUF_FACTOR, 0)); cov = np.array(b.read(L.BUF_COV, 0))
    F, _, _ = orc.mvn_factor_svd(cov, "harmonic")
    Gm = A @ A.T; dd = np.sqrt(np.diag(Gm)); orth = np.abs(Gm / dd[:, None] / dd[None, :] - np.eye(len(dd))).max()
    print("N %d tol 1e-%d iteration %d: %d sweeps, sample diff vs LAPACK %.3g px, recon %.2g, max |cos| %.2g, n_obs %d"
          % (N, L.get_option("oj_tol_exp"), it, int(s.lml), np.abs(Z @ A - Z @ F).max() * s.y_s, np.abs(A.T @ A - cov).max() / np.abs(cov).max(), orth, s.n_obs), flush=True)
ms = b.profile_stage(1, 3)
print("N %d tol 1e-%d: factor (warm) %.2f ms" % (N, L.get_option("oj_tol_exp"), ms), flush=True)
