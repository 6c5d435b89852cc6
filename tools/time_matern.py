import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    import gaussian_process_edge_trace_amd as pkg
    from oracle import gpet_oracle as orc
    ctx = pkg._lib.Context(0)
    img, truth = orc.synth_sinusoid_image(N, 5)
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    warm = truth[16:-16:16][:, [1, 0]].astype(np.int64)
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1,
              N_samples=300, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, seed=3, fix_endpoints=True)
    tr = pkg.GP_Edge_Tracing(init, grad, obs=warm, **kw, _ctx=ctx)
    b = tr._batch
    b.set_obs(0, warm)
    t = time.time(); b.iterate([3], 1); dt = time.time() - t
    s = b.scalars()
    print("N=%d first iteration %.2f s rank=%d sweeps=%d n=%d" % (N, dt, s.rank, int(s.lml), s.n), flush=True)
    for i, name in enumerate(["fit_predict_cov", "factor", "normals", "gemm", "score", "kde"]):
        print(name, "%.2f ms" % b.profile_stage(i, 1), flush=True)
    t = time.time(); et = tr(); print("full trace %.2f s, iters %d" % (time.time() - t, tr._n_iter))
    t = time.time(); orc.trace(init, grad, obs=warm, sign_convention="harmonic", **kw); print("oracle trace %.2f s" % (time.time() - t))


if __name__ == "__main__":
    main()
