import sys; sys.path.insert(0,'.')
import numpy as np
import gaussian_process_edge_trace_amd as amd
L=amd._lib; ctx=L.Context(0)
N=128
MKW = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 20, 'length_scale': 6}, noise_y=1, N_samples=200, score_thresh=1, delta_x=6, keep_ratio=0.1, pixel_thresh=3, fix_endpoints=True)
img, edge = amd.gpet_utils.construct_test_img((N, N), int(0.4 * N), 4, 0.05, 'sinusoidal', 0.3, gaps=True, seed=20)
init = edge[[0, -1], :][:, [1, 0]]
frames = [amd.gpet_utils.comp_grad_img(amd.gpet_utils.construct_test_img((N, N), int(0.4 * N * (1 + 0.02 * t)), 4, 0.05, 'sinusoidal', 0.3, gaps=True, seed=20 + t)[0], amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx) for t in range(6)]
for nch, fr in ((2, frames[:4]), (3, frames)):
    st = amd.SequenceTracer(fr, init, n_chains=nch, warm_every=12, seed=5, _ctx=ctx, **MKW)
    try:
        out = st()
        print("chains", nch, "ok iterations", st.iterations)
    except Exception as e:
        print("chains", nch, "FAILED", e)
        b = st._tracer._batch
        for e_ in range(b.B):
            s = b.scalars(e_); print("  edge", e_, "rank", s.rank, "n", s.n, "n_obs", s.n_obs, "status", s.status, "iter", s.iter, "done", s.done)
            cov = b.read(L.BUF_COV, e_); w = np.linalg.eigvalsh(cov); print("  cov eig min/max", w[0], w[-1], "diag min/max", np.diag(cov).min(), np.diag(cov).max(), "n diag < 1e-14*max:", int((np.diag(cov) < 1e-14*np.diag(cov).max()).sum()))
