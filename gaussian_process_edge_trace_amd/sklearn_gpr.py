"""``GaussianProcessRegressor`` / ``WeightedWhiteKernel`` mirrors of ``gp_edge_tracing/sklearn_gpr.py``.

The reference vendors a modified scikit-learn GPR (sklearn_gpr.py:31-610) and uses it only inside
``GP_Edge_Tracing.fit_predict_GP``.  This mirror keeps the parts of that interface the edge tracer
exercises -- ``fit`` with ``optimizer=None`` and ``predict`` (mean / std) -- with the reference's
modified ``normalize_y`` semantics (True centres only, False standardises; sklearn_gpr.py:221-234)
and runs them in libgpet_hip.so (the kernels of the converged fit: Cholesky + alpha + forward
substitution per query point).  Query points must be an arithmetic progression (the tracer's
x-grid).  ``predict(return_cov=True)``, ``sample_y`` (the device eigen-factor sampler on the query grid)
and ``log_marginal_likelihood`` run on the device too; hyper-parameter optimisation (``optimizer != None``)
lives in ``GP_Edge_Tracing`` (the converged fit), not here.
"""
from __future__ import annotations

import numpy as np

from . import _lib


class WeightedWhiteKernel(object):
    """Per-observation white noise ``noise_level * noise_weight[i]`` on the training diagonal
    (sklearn_gpr.py:617-721): zero for query points, zero when the row count equals ``edge_length``."""

    def __init__(self, noise_weight, edge_length, noise_level=1.0, noise_level_bounds=(1e-5, 1e5)):
        self.noise_weight = np.asarray(noise_weight, dtype=np.float64)
        self.edge_length = int(edge_length)
        self.noise_level = float(noise_level)
        self.noise_level_bounds = noise_level_bounds


def _unpack_kernel(kernel):
    """(kernel_type, nu, constant, length_scale, white) from ``C * RBF|Matern [+ WeightedWhiteKernel]``
    given as scikit-learn kernel objects (gpet.py:165-178,253) or as a dict
    {'kernel', 'constant', 'length_scale'[, 'nu'][, 'white']}."""
    if isinstance(kernel, dict):
        return (kernel["kernel"], float(kernel.get("nu", 2.5)), float(kernel["constant"]),
                float(kernel["length_scale"]), kernel.get("white"))
    white = None
    k = kernel
    if hasattr(k, "k1") and isinstance(getattr(k, "k2", None), WeightedWhiteKernel):
        white, k = k.k2, k.k1
    const, base = k.k1, k.k2
    name = type(base).__name__
    if name not in ("RBF", "Matern"):
        raise ValueError("kernel must be ConstantKernel * RBF or ConstantKernel * Matern")
    return name, float(getattr(base, "nu", 2.5)), float(const.constant_value), float(base.length_scale), white


class _Sum(object):
    """``product_kernel + WeightedWhiteKernel`` when the left operand is a scikit-learn kernel."""

    def __init__(self, k1, k2):
        self.k1, self.k2 = k1, k2


def add_white(kernel, white):
    """``kernel + white`` (scikit-learn's ``+`` would reject the mirror class)."""
    return _Sum(kernel, white)


class GaussianProcessRegressor(object):
    def __init__(self, kernel=None, alpha=1e-10, optimizer=None, n_restarts_optimizer=0, normalize_y=False,
                 copy_X_train=True, random_state=None, *, device=0, _ctx=None):
        if optimizer is not None:
            raise NotImplementedError("hyper-parameter optimisation runs inside GP_Edge_Tracing.__call__")
        self.kernel = kernel
        self.alpha = alpha
        self.optimizer = optimizer
        self.n_restarts_optimizer = n_restarts_optimizer
        self.normalize_y = normalize_y
        self.copy_X_train = copy_X_train
        self.random_state = random_state
        self._ctx = _ctx if _ctx is not None else _lib.Context(device)

    def fit(self, X, y):
        X = np.asarray(X, dtype=np.float64).reshape(len(y), -1)
        if X.shape[1] != 1:
            raise ValueError("1-D inputs only (the edge tracer's x coordinate)")
        y = np.asarray(y, dtype=np.float64)
        kt, nu, const, ell, white = _unpack_kernel(self.kernel)
        n = X.shape[0]
        self._y_train_mean = float(np.mean(y))
        sd = float(np.std(y))
        self._y_train_std = 1.0 if sd == 0.0 else sd  # _handle_zeros_in_scale, scalar path
        # the reference's swapped semantics: True centres only, False standardises
        yt = (y - self._y_train_mean) if self.normalize_y else (y - self._y_train_mean) / self._y_train_std
        w = np.zeros(n)
        nl = 0.0
        if white is not None:
            nl = white.noise_level
            w = np.asarray(white.noise_weight, dtype=np.float64)
            if w.shape[0] != n:
                raise ValueError("X must have the same length as weight ({:d}!={:d})".format(n, w.shape[0]))
            if n == white.edge_length:
                w = np.zeros(n)
        alpha = np.broadcast_to(np.asarray(self.alpha, dtype=np.float64), (n,))
        if n > 4096:
            raise NotImplementedError("more than 4096 training points")
        self._fit = dict(kt=kt, nu=nu, const=const, ell=ell, x=X[:, 0].copy(), yt=yt, noise=nl * w + alpha)
        self.X_train_, self.y_train_ = X, yt
        return self

    def _device_fit_predict(self, X, n_samples=1, want_factor=False):
        """One throw-away batch of a single edge whose grid is 0..Lq-1 ((x - X_m)/X_s maps it onto the query points):
        the converged-fit kernels evaluate K, its Cholesky factor, alpha and the posterior at the queries."""
        f = self._fit
        xq = np.asarray(X, dtype=np.float64).reshape(-1)
        Lq = xq.shape[0]
        step = (xq[-1] - xq[0]) / (Lq - 1) if Lq > 1 else 1.0
        if Lq < 4 or step <= 0 or not np.allclose(xq, xq[0] + step * np.arange(Lq), rtol=0, atol=1e-9 * max(1.0, abs(step))):
            raise NotImplementedError("query points must be an increasing arithmetic progression of >= 4 points")
        p = _lib.GpetParams()
        p.kernel_type = _lib.KERNEL_MATERN if f["kt"] == "Matern" else _lib.KERNEL_RBF
        p.nu, p.sigma_f, p.length_scale, p.noise_y = f["nu"], 1.0, f["ell"], 1.0
        p.n_samples, p.n_keep, p.delta_x, p.pixel_thresh, p.score_thresh = max(1, int(n_samples)), 1, max(2, Lq // 4), 2, 1.0
        p.fix_endpoints, p.x_st, p.x_en, p.n_init = 1, 0, Lq - 1, 1
        p.obs_cap, p.jitter = max(8, len(f["x"])), 0.0
        p.factor_cap, p.z_cols = (Lq, Lq) if want_factor else (4, 4)
        img = np.zeros((4, Lq), dtype=np.float32)
        img[0, 0] = 1.0
        b = _lib.Batch(self._ctx, [img], [p], [np.array([[0, 0]], dtype=np.int64)])
        # per-point noise folded into the weights: K_ii = const + 1.0 * noise_i + 1e-6 - 1e-6
        b.final_set_training_all([f["x"]], [f["yt"]], [f["noise"] - 1e-6])
        X_s = 1.0 / step
        par = np.zeros((1, 12))
        par[0, :9] = [f["const"], f["ell"], 1.0, -xq[0] * X_s, X_s, 0.0, 1.0,
                      self._y_train_mean, self._y_train_std]
        mean, std = b.final_predict_all(par)
        return b, mean[0, :Lq].copy(), std[0, :Lq].copy()

    def predict(self, X, return_std=False, return_cov=False):
        """Posterior mean [, std | covariance] at an arithmetic progression of query points (sklearn_gpr.py:323-438)."""
        if return_std and return_cov:
            raise RuntimeError("At most one of return_std or return_cov can be requested.")  # sklearn_gpr.py:355-357
        b, mean, std = self._device_fit_predict(X)
        try:
            if return_cov:
                b.final_cov()  # (K** - V^T V) * y_train_std^2 on the f64 matrix cores
                return mean, b.read(_lib.BUF_COV)
        finally:
            b.close()
        if return_std:
            return mean, std
        return mean

    def sample_y(self, X, n_samples=1, random_state=0):
        """``n_samples`` posterior curves at the query points, (N, n_samples), drawn like the reference draws them
        (sklearn_gpr.py:440-473 -> numpy legacy ``multivariate_normal``): the RandomState(random_state) normal stream
        times the sqrt(s) v rows of the covariance's symmetric SVD, all on the device.  Eigenvector signs follow the
        library's convention (include/gpet_hip.h, gpet_gp_factor); LAPACK's are implementation-defined."""
        n_samples = int(n_samples)
        b, mean, _ = self._device_fit_predict(X, n_samples, want_factor=True)
        try:
            b.final_cov()
            b.factor()
            b.normals([int(random_state)])
            sc = b.scalars()
            sc.y_s = 1.0
            b.write_scalars(sc)
            b.write(_lib.BUF_MEAN, mean)
            b.sample()
            return b.read(_lib.BUF_SAMPLES).T.copy()
        finally:
            b.close()

    def log_marginal_likelihood(self, theta=None, eval_gradient=False):
        """Log marginal likelihood of theta = log(constant, length_scale, noise_level) on the training set of ``fit``
        [and its gradient] (sklearn_gpr.py:475-585), evaluated by the batched objective kernel of the converged fits;
        theta=None: at the kernel's own hyper-parameters."""
        f = self._fit
        if theta is None:
            theta = np.log([f["const"], f["ell"], 1.0])
        b, _, _ = self._device_fit_predict(np.arange(4.0))
        try:
            val, grad = b.lml_batch([0], np.asarray(theta, dtype=np.float64).reshape(1, 3))
        finally:
            b.close()
        if eval_gradient:
            return -float(val[0]), -grad[0]
        return -float(val[0])
