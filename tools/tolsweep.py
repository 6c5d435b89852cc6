import sys; sys.path.insert(0,'.')
import numpy as np
import gaussian_process_edge_trace_amd as amd
from oracle import gpet_oracle as orc
L=amd._lib; ctx=L.Context(0)
N=1024
img, truth = orc.synth_sinusoid_image(N, 5)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
init = truth[[0, -1], :][:, [1, 0]]
kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1, N_samples=300, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, seed=3, fix_endpoints=True)
Z = orc.legacy_standard_normal(11, 64 * 1024).reshape(64, 1024)
for step in (16, 64, 0):
    warm = truth[step:-step:step][:, [1, 0]].astype(np.int64) if step else np.zeros((0,2),np.int64)
    tr = amd.GP_Edge_Tracing(init, grad, obs=warm, **kw, _ctx=ctx); b = tr._batch
    b.set_obs(0, warm); b.fit_predict(True)
    cov=b.read(L.BUF_COV); F,_,_=orc.mvn_factor_svd(cov,"harmonic")
    for ex in (11, 9, 8, 7, 6, 5):
        L.set_option("oj_tol_exp", ex)
        b.factor(); s=b.scalars(); A=b.read(L.BUF_FACTOR)
        ms=b.profile_stage(1,2)
        Gm = A @ A.T; dd = np.sqrt(np.diag(Gm)); orth = np.abs(Gm / dd[:, None] / dd[None, :] - np.eye(len(dd))).max()
        print("obs step %d tol 1e-%d: %d sweeps, %.1f ms, max sample diff vs LAPACK %.3g px, recon %.2g, max |cos(row_p, row_q)| %.2g" % (step, ex, int(s.lml), ms, np.abs(Z@A-Z@F).max()*s.y_s, np.abs(A.T@A-cov).max()/np.abs(cov).max(), orth), flush=True)
    L.set_option("oj_tol_exp", 11)
