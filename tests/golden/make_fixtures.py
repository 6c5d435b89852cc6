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

# The signs of the reference's SVD rows.  numpy's legacy multivariate_normal (sklearn_gpr.py:464) looks numpy.linalg.svd up when
# it is called, so a recorder in its place sees every factorisation of a trace: per call the bit "sum_j row_k[j] / (j + 1) >= 0"
# of every row sqrt(s_k) v_k -- the library fixes that sum non-negative (include/gpet_hip.h, gpet_gp_factor), so these bits are
# exactly what turns the device's rows into LAPACK's: a device test flips the N(0, 1) column k where the bit is 0 and must then
# reproduce the reference's own samples, observation sets and trace.
_svd_bits = None
_orig_svd = np.linalg.svd


def _svd_hook(a, *args, **kw):
    out = _orig_svd(a, *args, **kw)
    if _svd_bits is not None and getattr(a, "ndim", 0) == 2 and a.shape[0] == a.shape[1] and not args and not kw:
        _, sv, vt = out
        rows = np.sqrt(sv)[:, None] * vt
        _svd_bits.append((rows @ (1.0 / np.arange(1, rows.shape[1] + 1)) >= 0))
    return out


np.linalg.svd = _svd_hook


def make_image(N, seed):
    img, edge = orc.synth_sinusoid_image(N, seed)
    kern = gpet_utils.kernel_builder(size=(11, 5), unit=False)
    grad = gpet_utils.comp_grad_img(img, kern)
    return img, edge, kern, grad


def stage_fixture(name, N, img_seed, ctor_kw, obs, gp_seed, keep_samples=None, keep_factor_rows=None, light=False, pixels=True):
    """``light`` (images above 500 px): nothing of image size is stored -- the image is the oracle generator's
    (``in_img_seed``; the device's comp_grad_img of it equals the reference's bit for bit, tests/test_gpu_stages.py) and the
    training-set matrices K / L (n^2) are left out; ``pixels=False`` skips the pixel selection (more observations than
    algo_thresh: the reference's loop would not run it)."""
    global _svd_bits
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
    if light:
        for k in ("ref_grad", "ref_grad64", "ref_grad_kde"):
            out.pop(k)
        out["ref_grad_probe"] = grad[::97, ::89].copy()      # (a few hundred pixels of each: the regenerated images must hit them)
        out["ref_grad_kde_probe"] = tr.grad_kde[::97, ::89].astype(np.float32)
    # ---- a2..a6: one not-converged GP iteration
    _captured.clear()
    _svd_bits = []
    Y = tr.fit_predict_GP(obs, converged=False, seed=gp_seed)
    out["ref_svd_sign_bits"] = np.packbits(_svd_bits[-1])  # (of the factorisation sample_y itself made)
    _svd_bits = None
    gp = _captured[-1]
    xg = tr.x_grid[:, None].astype(np.float64)
    mean, cov = gp.predict(xg, return_cov=True)
    _, std = gp.predict(xg, return_std=True)
    _, s, v = np.linalg.svd(cov)
    factor = np.sqrt(s)[:, None] * v
    Z = np.random.RandomState(gp_seed).standard_normal((tr.N_samples, tr.edge_length))
    K = gp._K_train_dbg.copy()
    K[np.diag_indices_from(K)] += gp.alpha
    if light:
        out.update(ref_L_diag=np.diag(gp.L_).copy(), ref_L_lastrow=gp.L_[-1].copy())
    else:
        out.update(ref_K=K, ref_L=gp.L_)
    out.update(ref_X_train=gp.X_train_[:, 0], ref_y_train=gp.y_train_,
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
    if keep_samples is not None and not light:
        # the fixture keeps only a few of the samples: store the kept curves themselves (y only; x = the grid), so that
        # the curve-KDE stage can be fed through the get_best_pixels(curves, costs, pre_fobs) seam
        assert np.array_equal(bc[:, :, 0], np.repeat(tr.x_grid[:, None], bc.shape[1], axis=1))
        out["ref_best_curves_y"] = bc[:, :, 1].copy()
    # ---- f1: KDE stand-in + pixel selection (reference compute_new_obs on stored inputs)
    if pixels:
        kde_arr = tr.kernel_density_estimate(bc, bcost)
        pix = np.argwhere(kde_arr > tr.kde_thresh)
        if tr.fix_endpoints:
            pix = pix[(pix[:, 1] > tr.x_st) & (pix[:, 1] < tr.x_en)]
        thresh_in = tr.score_thresh
        fobs = tr.compute_new_obs(pix, kde_arr, obs[:, [1, 0]].reshape(-1, 2))
        out.update(in_score_thresh=thresh_in, ref_score_thresh_out=tr.score_thresh, ref_fobs=fobs)
        if light:
            out["ref_kde_probe"] = kde_arr[::97, ::89].astype(np.float32)
        else:
            out["ref_kde_arr"] = kde_arr.astype(np.float32)
    # ---- ctor-derived scalars
    out.update(ref_scalars=np.array([tr.x_st, tr.x_en, tr.N_samples, tr.N_keep, tr.N_subints, tr.algo_thresh,
                                     tr.delta_x, tr.pixel_thresh, tr.edge_length], dtype=np.int64),
               ref_sigma=np.array([tr.sigma_f, tr.sigma_l, tr.kernel_nu, tr.keep_ratio], dtype=np.float64),
               ref_kernel_type=np.array(tr.kernel_type))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith("ref_")})


def trace_fixture(name, N, img_seed, ctor_kw):
    global _svd_bits
    img, edge, kern, grad = make_image(N, img_seed)
    init = edge[[0, -1], :][:, [1, 0]]
    kw = dict(ctor_kw)
    kw["return_std"] = False
    tr = gpet.GP_Edge_Tracing(init, grad, **kw)
    _captured.clear()
    _svd_bits = []
    et, (all_samples, all_obs, iter_curves) = tr(return_lines=True)
    bits, _svd_bits = _svd_bits, None
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
    assert len(bits) == len(all_obs) - 2, (len(bits), len(all_obs))  # one factorisation per loop iteration (all_obs: start + per iteration + the final append)
    for i, b_ in enumerate(bits):
        out["ref_svd_sign_bits_%02d" % i] = np.packbits(b_)  # iteration i: observation set ref_obs_i -> ref_obs_(i+1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "iters", out["ref_n_iter"], "edge", et.shape)


def _replay_one(args):
    """(image seed, RNG seed, ctor kwargs, N) -> what a replay needs of one reference trace: the observation set after every
    iteration, the SVD sign bits of every iteration, the final trace."""
    global _svd_bits
    img_seed, seed, kw, N = args
    img, edge, kern, grad = make_image(N, img_seed)
    init = edge[[0, -1], :][:, [1, 0]]
    kw = dict(kw, seed=seed, return_std=False)
    tr = gpet.GP_Edge_Tracing(init, grad, **kw)
    _svd_bits = []
    et, (all_samples, all_obs, iter_curves) = tr(return_lines=True)
    bits, _svd_bits = _svd_bits, None
    n_iter = len(all_obs) - 2
    assert len(bits) == n_iter
    obs = [np.asarray(o).reshape(-1, 2).astype(np.int16) for o in all_obs[:-1]]
    return dict(img_seed=img_seed, seed=seed, n_iter=n_iter, edge_trace=et.astype(np.int16), obs=obs,
                bits=[np.packbits(b_) for b_ in bits])


def replay_fixture(name, N, kw, img_seeds, seeds, workers=4):
    """Many reference traces of one configuration, each reduced to what the GPU replay needs (observation sets as int16, one sign
    bit per factor row and iteration, the final trace): ~10 KB per trace."""
    import multiprocessing as mp
    jobs = [(a, b, kw, N) for a in img_seeds for b in seeds]
    with mp.get_context("fork").Pool(workers) as pool:
        rows = pool.map(_replay_one, jobs, chunksize=1)
    out = dict(in_N=N, in_img_seeds=np.array([r["img_seed"] for r in rows]), in_seeds=np.array([r["seed"] for r in rows]),
               ref_n_iter=np.array([r["n_iter"] for r in rows]), kde_standin=1)
    for t, r in enumerate(rows):
        out["ref_edge_trace_%02d" % t] = r["edge_trace"]
        out["ref_obs_counts_%02d" % t] = np.array([o.shape[0] for o in r["obs"]], dtype=np.int16)
        out["ref_obs_all_%02d" % t] = np.concatenate(r["obs"], axis=0) if sum(o.shape[0] for o in r["obs"]) else np.zeros((0, 2), np.int16)
        out["ref_svd_sign_bits_%02d" % t] = np.stack(r["bits"], axis=0)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, len(rows), "traces, iterations", out["ref_n_iter"].tolist())


def readme_fixture(name="readme_trace"):
    """The README's demo (README.md:46-85) AS WRITTEN on the reference's own image (tests/golden/readme_image.npz, made by the
    unmodified construct_test_img under scikit-image 0.18.3: make_readme_image.py): the positional call binds seed = 1 to
    pixel_thresh (clamped to 2), return_std = True to seed (= 1) and fix_endpoints to return_std (quirk Q6)."""
    global _svd_bits
    ri = np.load(os.path.join(HERE, "readme_image.npz"))
    test_img, true_edge = ri["ref_img"], ri["ref_true_edge"]
    kernel = gpet_utils.kernel_builder(size=(11, 5), unit=False)
    grad_img = gpet_utils.comp_grad_img(test_img, kernel)
    assert np.array_equal(grad_img, ri["ref_grad_py39"])  # scipy 1.7.1 (python3.9) and 1.15.3 (here) convolve alike
    kernel_params = {'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}
    delta_x, score_thresh, N_samples, noise_y, seed, keep_ratio = 5, 1, 1000, 1, 1, 0.1
    init = true_edge[[0, -1], :][:, [1, 0]]
    obs = np.array([])
    fix_endpoints, return_std = True, True
    noisy_trace = gpet.GP_Edge_Tracing(init, grad_img, kernel_params, noise_y, obs, N_samples, score_thresh,
                                       delta_x, keep_ratio, seed, return_std, fix_endpoints)
    _captured.clear()
    edge_pred, edge_credint = noisy_trace(False, False, False, False)
    gp = _captured[-1]
    again = gpet.GP_Edge_Tracing(init, grad_img, kernel_params, noise_y, obs, N_samples, score_thresh,
                                 delta_x, keep_ratio, seed, return_std, fix_endpoints)
    # (return_std is True in this binding, so return_lines is not honoured, gpet.py:902-908: the observation sets are recorded
    #  where the loop receives them, at the get_best_pixels seam of the instance)
    all_obs = [np.zeros((0, 2), dtype=np.int64)]
    seam = again.get_best_pixels

    def recording_seam(*a, **k):
        fobs = seam(*a, **k)
        all_obs.append(np.asarray(fobs).reshape(-1, 2).copy())
        return fobs
    again.get_best_pixels = recording_seam
    _svd_bits = []
    et2, ci2 = again(False, False, False, False)
    bits, _svd_bits = _svd_bits, None
    assert np.array_equal(et2, edge_pred)
    out = dict(in_init=init, ref_edge_trace=edge_pred, ref_ci_lower=edge_credint[0], ref_ci_upper=edge_credint[1],
               ref_n_iter=len(all_obs) - 1, ref_final_theta=gp.kernel_.theta, ref_final_lml=gp.log_marginal_likelihood_value_,
               ref_bound=np.array([noisy_trace.pixel_thresh, noisy_trace.seed, int(noisy_trace.return_std), int(noisy_trace.fix_endpoints),
                                   noisy_trace.algo_thresh], dtype=np.int64),
               ref_metrics=np.array([gpet_utils.trace_MSE(edge_pred, true_edge), gpet_utils.trace_relarea(edge_pred, true_edge),
                                     gpet_utils.trace_dicecoef(edge_pred, true_edge)], dtype=np.float64),
               published_metrics=np.array([12.604, 0.00339, 0.9953]),  # Figures/noisy_trace_results.png (real KDEpy)
               kde_standin=1)
    for i, o in enumerate(all_obs):
        out["ref_obs_%02d" % i] = np.asarray(o).reshape(-1, 2)
    for i, b_ in enumerate(bits):
        out["ref_svd_sign_bits_%02d" % i] = np.packbits(b_)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "iterations", out["ref_n_iter"], "bound (pixel_thresh, seed, return_std, fix_endpoints, algo_thresh)", out["ref_bound"],
          "MSE / rel. area / DICE", out["ref_metrics"], "published", out["published_metrics"])


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
    if want("readme_trace"):
        readme_fixture()
    if want("replay_rbf500"):
        # twelve more reference traces of the README configuration (pixel_thresh = 5): two images x six RNG seeds 997 apart
        replay_fixture("replay_rbf500", 500, dict(readme), (1, 3), [1000 + 997 * k for k in range(6)])
    if want("replay_default500"):
        # the reference's DEFAULT kernel (kernel_options = (1, 3, 3): Matern-5/2, sigma_f = M // 6, l = edge_length // 2,
        # gpet.py:25,139-151) at the README's size: full-rank covariances, the any-rank factor of the device
        dflt = dict(readme)
        dflt["kernel_options"] = (1, 3, 3)
        replay_fixture("replay_default500", 500, dflt, (1,), [1, 998], workers=2)
    if want("stage_rbf2048_n1500"):
        # BASELINE config 3's shape (tests/test_gpu_configs.py::test_config3_large_n_gp_iteration builds the same inputs):
        # 2048^2 image, 1498 user-supplied observations (+ 2 inits = 1500 training points), N_samples = 4000
        N3 = 2048
        _, edge3 = orc.synth_sinusoid_image(N3, 0)
        rng3 = np.random.default_rng(0)
        cols3 = np.sort(rng3.choice(np.arange(1, N3 - 1), size=1498, replace=False))
        obs3 = np.stack([cols3, edge3[cols3, 0] + rng3.integers(-2, 3, size=cols3.size)], axis=1).astype(np.int64)
        stage_fixture("stage_rbf2048_n1500", N3, 0, dict(kernel_options={'kernel': 'RBF', 'sigma_f': 300, 'length_scale': 80}, noise_y=1,
                                                         N_samples=4000, score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5,
                                                         seed=1, fix_endpoints=True),
                      obs3, 7, keep_samples=48, keep_factor_rows=96, light=True, pixels=False)
    if want("stage_mat1024"):
        # BASELINE config 5's frame shape: 1024^2, Matern-5/2 sigma_f = 154, l = 41, warm start = every 16th pixel of the previous
        # (here: the true) trace -- a full-rank covariance, the any-rank factor
        N5 = 1024
        _, edge5 = orc.synth_sinusoid_image(N5, 100)
        obs5 = edge5[16:-16:16][:, [1, 0]].astype(np.int64)
        stage_fixture("stage_mat1024", N5, 100, dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 154, 'length_scale': 41},
                                                     noise_y=1, N_samples=1000, score_thresh=1, delta_x=8, keep_ratio=0.1,
                                                     pixel_thresh=5, seed=3, fix_endpoints=True),
                      obs5, 9, keep_samples=48, keep_factor_rows=128, light=True, pixels=True)
