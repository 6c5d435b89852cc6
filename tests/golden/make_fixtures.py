#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference
(/root/reference, imported through ref_harness.py's shims).  Runs only in the build
container; the resulting small .npz files are committed and travel to the GPU box.

    python tests/golden/make_fixtures.py

Every array named ``ref_*`` was produced by the reference's own code (plus the real
third-party numerics of this image: numpy 2.2.6 legacy RNG + LAPACK SVD, scipy 1.15.3,
scikit-learn 1.7.2 kernels).  Arrays named ``in_*`` are inputs.  Anything downstream of the
KDE carries ``kde_standin=1`` because KDEpy is absent.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")

import ref_harness  # noqa: E402

ref = ref_harness.load_reference()
from gp_edge_tracing import gpet, gpet_utils, sklearn_gpr  # noqa: E402

from oracle import gpet_oracle as orc  # noqa: E402  (only for the synthetic image recipe)

_captured = []
_orig_fit = sklearn_gpr.GaussianProcessRegressor.fit


def _fit_hook(self, X, y):
    out = _orig_fit(self, X, y)
    # K must be captured now: WeightedWhiteKernel zeroes its weights on the first
    # edge_length-row call (sklearn_gpr.py:673-677), i.e. during predict.
    self._K_train_dbg = self.kernel_(self.X_train_)
    _captured.append(self)
    return out


sklearn_gpr.GaussianProcessRegressor.fit = _fit_hook


def make_image(N, seed):
    img, edge = orc.synth_sinusoid_image(N, seed)
    kern = gpet_utils.kernel_builder(size=(11, 5), unit=False)
    grad = gpet_utils.comp_grad_img(img, kern)
    return img, edge, kern, grad


def stage_fixture(name, N, img_seed, ctor_kw, obs, gp_seed, keep_samples=None, keep_factor_rows=None):
    img, edge, kern, grad = make_image(N, img_seed)
    init = edge[[0, -1], :][:, [1, 0]]
    if ctor_kw.pop("_swap_init", False):
        init = init[::-1].copy()
    tr = gpet.GP_Edge_Tracing(init, grad, **ctor_kw)
    # grad64 / grad_kde / kde_arr hold float32-representable values (normalise() rounds through
    # float32, gpet_utils.py:81-91), so they are stored as float32 without loss.
    assert np.array_equal(tr.grad_img, tr.grad_img.astype(np.float32).astype(np.float64))
    assert np.array_equal(tr.grad_kde, tr.grad_kde.astype(np.float32).astype(np.float64))
    out = dict(in_kernel=kern, ref_grad=grad, in_init=init, in_obs=obs, in_gp_seed=gp_seed, in_img_seed=img_seed,
               in_true_edge=edge, ref_grad64=tr.grad_img.astype(np.float32),
               ref_grad_kde=tr.grad_kde.astype(np.float32), kde_standin=1)
    if N <= 128:
        out["in_img"] = img
    # ---- a2..a6: one not-converged GP iteration
    _captured.clear()
    Y = tr.fit_predict_GP(obs, converged=False, seed=gp_seed)
    gp = _captured[-1]
    xg = tr.x_grid[:, None].astype(np.float64)
    mean, cov = gp.predict(xg, return_cov=True)
    _, std = gp.predict(xg, return_std=True)
    _, s, v = np.linalg.svd(cov)
    factor = np.sqrt(s)[:, None] * v
    Z = np.random.RandomState(gp_seed).standard_normal((tr.N_samples, tr.edge_length))
    K = gp._K_train_dbg.copy()
    K[np.diag_indices_from(K)] += gp.alpha
    out.update(ref_X_train=gp.X_train_[:, 0], ref_y_train=gp.y_train_, ref_K=K, ref_L=gp.L_,
               ref_alpha=gp.alpha_, ref_y_train_mean=gp._y_train_mean, ref_y_train_std=gp._y_train_std,
               ref_amp=tr.constant_kernel.constant_value, ref_mean=mean, ref_std=std, ref_svals=s,
               ref_cov_diag=np.diag(cov).copy(), ref_cov_row0=cov[0].copy(), ref_cov_rowmid=cov[cov.shape[0] // 2].copy())
    if keep_factor_rows is None:
        out["ref_factor"] = factor
        out["ref_cov"] = cov
    else:
        out["ref_factor_top"] = factor[:keep_factor_rows]
    ns = Y.shape[1] if keep_samples is None else keep_samples
    out["ref_samples_head"] = Y[:, :ns]
    out["ref_Z_head"] = Z[:min(8, Z.shape[0])]
    # ---- a7: scoring on the reference's own samples
    bc, bcost, (oc, ocost) = tr.get_best_curves(Y)
    costs = np.asarray([tr.cost_funct(np.stack((tr.x_grid, Y[:, i]), axis=-1)) for i in range(Y.shape[1])])
    out.update(ref_costs=costs, ref_best_idxs=np.argsort(costs)[:tr.N_keep], ref_best_costs=bcost,
               ref_best_curve0=bc[:, 0, :])
    if keep_samples is not None:
        # the fixture keeps only a few of the samples: store the kept curves themselves (y only; x = the grid), so that
        # the curve-KDE stage can be fed through the get_best_pixels(curves, costs, pre_fobs) seam
        assert np.array_equal(bc[:, :, 0], np.repeat(tr.x_grid[:, None], bc.shape[1], axis=1))
        out["ref_best_curves_y"] = bc[:, :, 1].copy()
    # ---- f1: KDE stand-in + pixel selection (reference compute_new_obs on stored inputs)
    kde_arr = tr.kernel_density_estimate(bc, bcost)
    pix = np.argwhere(kde_arr > tr.kde_thresh)
    if tr.fix_endpoints:
        pix = pix[(pix[:, 1] > tr.x_st) & (pix[:, 1] < tr.x_en)]
    thresh_in = tr.score_thresh
    fobs = tr.compute_new_obs(pix, kde_arr, obs[:, [1, 0]].reshape(-1, 2))
    out.update(ref_kde_arr=kde_arr.astype(np.float32), in_score_thresh=thresh_in,
               ref_score_thresh_out=tr.score_thresh, ref_fobs=fobs)
    # ---- ctor-derived scalars
    out.update(ref_scalars=np.array([tr.x_st, tr.x_en, tr.N_samples, tr.N_keep, tr.N_subints, tr.algo_thresh,
                                     tr.delta_x, tr.pixel_thresh, tr.edge_length], dtype=np.int64),
               ref_sigma=np.array([tr.sigma_f, tr.sigma_l, tr.kernel_nu, tr.keep_ratio], dtype=np.float64),
               ref_kernel_type=np.array(tr.kernel_type))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith("ref_")})


def trace_fixture(name, N, img_seed, ctor_kw):
    img, edge, kern, grad = make_image(N, img_seed)
    init = edge[[0, -1], :][:, [1, 0]]
    kw = dict(ctor_kw)
    kw["return_std"] = False
    tr = gpet.GP_Edge_Tracing(init, grad, **kw)
    _captured.clear()
    et, (all_samples, all_obs, iter_curves) = tr(return_lines=True)
    gp = _captured[-1]
    kw["return_std"] = True
    tr2 = gpet.GP_Edge_Tracing(init, grad, **kw)
    et2, ci = tr2()
    assert np.array_equal(et, et2)
    out = dict(in_kernel=kern, in_init=init, in_true_edge=edge, in_img_seed=img_seed, ref_edge_trace=et,
               ref_ci_lower=ci[0], ref_ci_upper=ci[1], ref_final_mean=all_samples[-1],
               ref_n_iter=len(all_obs) - 2, ref_final_theta=gp.kernel_.theta,
               ref_final_lml=gp.log_marginal_likelihood_value_, kde_standin=1)
    if N <= 128:
        out["in_img"] = img
        out["ref_grad"] = grad
    else:
        out["in_grad_from"] = np.array("stage_rbf500.npz:ref_grad")
    for i, o in enumerate(all_obs[:-1]):
        out["ref_obs_%02d" % i] = np.asarray(o).reshape(-1, 2)
    for i, c in enumerate(iter_curves[:-1]):
        out["ref_optimal_curve_%02d" % i] = c[:, 1]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "iters", out["ref_n_iter"], "edge", et.shape)


def _quality_one(args):
    """(image seed, RNG seed) -> (n_iter, MSE, DICE, rel. area) of the UNMODIFIED reference on the README configuration."""
    img_seed, seed, kw = args
    img, edge, kern, grad = make_image(500, img_seed)
    init = edge[[0, -1], :][:, [1, 0]]
    kw = dict(kw, seed=seed, return_std=False)
    tr = gpet.GP_Edge_Tracing(init, grad, **kw)
    et, (all_samples, all_obs, iter_curves) = tr(return_lines=True)
    return (img_seed, seed, len(all_obs) - 2, float(gpet_utils.trace_MSE(et, edge)), float(gpet_utils.trace_dicecoef(et, edge)),
            float(gpet_utils.trace_relarea(et, edge)))


# RNG seeds of the quality fixture: 1..24 (consecutive: iteration i of seed s draws from RandomState(s + i + 1), so
# neighbouring seeds share most of their normal streams -- 24 strongly correlated runs) and 48 seeds 997 apart
QUALITY_SEEDS = list(range(1, 25)) + [1000 + 997 * k for k in range(48)]


def quality_fixture(name, kw, img_seeds=(1, 3), seeds=QUALITY_SEEDS, workers=4):
    """Trace quality of the reference itself over image seeds x RNG seeds (a few KB): the distribution the device's
    own quality is held against (T3: the device draws equally valid samples with other eigenvector signs, so single
    traces differ; their distribution must not)."""
    import multiprocessing as mp
    path = os.path.join(HERE, name + ".npz")
    have = np.load(path)["ref_quality"] if os.path.exists(path) else np.zeros((0, 6))
    done = {(int(r[0]), int(r[1])) for r in have}
    jobs = [(a, b, kw) for a in img_seeds for b in seeds if (a, b) not in done]  # (runs already stored are kept)
    with mp.get_context("fork").Pool(workers) as pool:
        rows = pool.map(_quality_one, jobs, chunksize=1)
    rows = np.asarray(sorted([tuple(r) for r in have] + rows), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), ref_quality=rows,
                        columns=np.array(["img_seed", "seed", "n_iter", "mse", "dice", "relarea"]), kde_standin=1)
    for a in img_seeds:
        r = rows[rows[:, 0] == a]
        print(name, "image seed", a, "n_iter", r[:, 2].min(), r[:, 2].max(), "MSE median %.1f max %.1f" % (np.median(r[:, 3]), r[:, 3].max()),
              "DICE median %.4f min %.4f" % (np.median(r[:, 4]), r[:, 4].min()))


def kernel_fixture():
    """Kernel matrices for every supported correlation family (sklearn kernels via the
    reference's own ctor path)."""
    from sklearn.gaussian_process import kernels as sk

    x = np.array([0.0, 3.0, 4.0, 10.0, 11.5, 40.0, 41.0, 63.0])
    xq = np.arange(0, 64, dtype=np.float64)
    out = dict(in_x=x, in_xq=xq)
    for tag, kern in [("rbf", sk.RBF(7.5)), ("m05", sk.Matern(7.5, nu=0.5)), ("m15", sk.Matern(7.5, nu=1.5)),
                      ("m25", sk.Matern(7.5, nu=2.5)), ("m35", sk.Matern(7.5, nu=3.5))]:
        k = sk.ConstantKernel(3.25, "fixed") * kern
        out["ref_Kxx_" + tag] = k(x[:, None])
        out["ref_Kqx_" + tag] = k(xq[:, None], x[:, None])
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **out)
    print("kernels")


if __name__ == "__main__":
    only = set(sys.argv[1:])  # optional: names of the fixtures to (re)generate

    def want(name):
        return not only or name in only

    if want("kernels"):
        kernel_fixture()
    rbf = {'kernel': 'RBF', 'sigma_f': 10, 'length_scale': 8}
    small = dict(kernel_options=rbf, noise_y=1, N_samples=128, score_thresh=1, delta_x=5, keep_ratio=0.1,
                 pixel_thresh=3, seed=1, fix_endpoints=True)
    if want("stage_rbf64"):
        stage_fixture("stage_rbf64", 64, 3, dict(small), np.array([[20, 40], [41, 25], [10, 30]], dtype=np.int64), 11)
    if want("stage_rbf65"):  # odd edge length: even Simpson sample count (scipy-version dependent rule)
        stage_fixture("stage_rbf65", 65, 3, dict(small), np.array([[20, 40], [41, 25], [10, 30]], dtype=np.int64), 11)
    if want("stage_mat128"):
        stage_fixture("stage_mat128", 128, 5, dict(kernel_options=(1, 3, 3), noise_y=0.5, N_samples=256,
                                                   score_thresh=0.9, delta_x=8, keep_ratio=0.125, pixel_thresh=4,
                                                   seed=7, fix_endpoints=False),
                      np.array([[30, 70], [64, 60], [100, 50], [90, 66]], dtype=np.int64), 23)
    if want("stage_mat15_96"):
        stage_fixture("stage_mat15_96", 96, 2, dict(kernel_options=(2, 2, 2), noise_y=1, N_samples=200, score_thresh=1,
                                                    delta_x=6, keep_ratio=0.1, pixel_thresh=2, seed=3,
                                                    fix_endpoints=True),
                      np.zeros((0, 2), dtype=np.int64), 4)
    mat35 = dict(kernel_options={'kernel': 'Matern', 'nu': 3.5, 'sigma_f': 15, 'length_scale': 10}, noise_y=1, N_samples=200,
                 score_thresh=1, delta_x=6, keep_ratio=0.1, pixel_thresh=3, seed=5, fix_endpoints=True)
    if want("stage_mat35_96"):  # general-nu Matern: the Bessel-K branch of sklearn's kernel (gpet.py:134)
        stage_fixture("stage_mat35_96", 96, 2, dict(mat35), np.array([[30, 50], [60, 40], [15, 52]], dtype=np.int64), 9)
    if want("trace_mat35_96"):
        trace_fixture("trace_mat35_96", 96, 2, dict(mat35))
    readme = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, noise_y=1, N_samples=1000,
                  score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, seed=1, fix_endpoints=True)
    if want("stage_rbf500"):
        # obs for the 500^2 stage fixture: the reference trace's own iteration-6 observation set
        img, edge, kern, grad = make_image(500, 1)
        tr = gpet.GP_Edge_Tracing(edge[[0, -1], :][:, [1, 0]], grad, **readme)
        _, (_, all_obs, _) = tr(return_lines=True)
        stage_fixture("stage_rbf500", 500, 1, dict(readme), np.asarray(all_obs[6]).reshape(-1, 2), 8,
                      keep_samples=48, keep_factor_rows=96)
    if want("trace_rbf64"):
        trace_fixture("trace_rbf64", 64, 3, dict(small))
    if want("trace_rbf65"):
        trace_fixture("trace_rbf65", 65, 3, dict(small))
    if want("trace_mat128"):
        trace_fixture("trace_mat128", 128, 5, dict(kernel_options=(1, 3, 3), noise_y=0.5, N_samples=256,
                                                   score_thresh=0.9, delta_x=8, keep_ratio=0.125, pixel_thresh=4,
                                                   seed=7, fix_endpoints=False))
    if want("trace_rbf500"):
        trace_fixture("trace_rbf500", 500, 1, dict(readme))
    if want("quality_rbf500"):
        quality_fixture("quality_rbf500", dict(readme))
