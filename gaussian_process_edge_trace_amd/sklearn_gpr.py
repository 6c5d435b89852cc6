"""``GaussianProcessRegressor`` / ``WeightedWhiteKernel`` mirrors of ``gp_edge_tracing/sklearn_gpr.py``.

The reference vendors a modified scikit-learn GPR (sklearn_gpr.py:31-610) and uses it only inside
``GP_Edge_Tracing.fit_predict_GP``.  This mirror keeps the parts of that interface the edge tracer
exercises -- ``fit`` with ``optimizer=None`` and ``predict`` (mean / std) -- with the reference's
modified ``normalize_y`` semantics (True centres only, False standardises; sklearn_gpr.py:221-234)
and runs them in libgpet_hip.so (the kernels of the converged fit: Cholesky + alpha + forward
substitution per query point).  Query points must be an arithmetic progression (the tracer's
x-grid).  Posterior draws go through ``GP_Edge_Tracing.fit_predict_GP`` (the sample path is built
around the pixel grid); ``sample_y`` and hyper-parameter optimisation are not offered here.
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

    def predict(self, X, return_std=False, return_cov=False):
        if return_cov:
            raise NotImplementedError("covariances are produced inside GP_Edge_Tracing.fit_predict_GP")
        f = self._fit
        xq = np.asarray(X, dtype=np.float64).reshape(-1)
        Lq = xq.shape[0]
        step = (xq[-1] - xq[0]) / (Lq - 1) if Lq > 1 else 1.0
        if Lq < 4 or step <= 0 or not np.allclose(xq, xq[0] + step * np.arange(Lq), rtol=0, atol=1e-9 * max(1.0, abs(step))):
            raise NotImplementedError("query points must be an increasing arithmetic progression of >= 4 points")
        # one throw-away batch of a single edge whose grid is 0..Lq-1; (x - X_m)/X_s maps it onto xq
        p = _lib.GpetParams()
        p.kernel_type = _lib.KERNEL_MATERN if f["kt"] == "Matern" else _lib.KERNEL_RBF
        p.nu, p.sigma_f, p.length_scale, p.noise_y = f["nu"], 1.0, f["ell"], 1.0
        p.n_samples, p.n_keep, p.delta_x, p.pixel_thresh, p.score_thresh = 1, 1, max(2, Lq // 4), 2, 1.0
        p.fix_endpoints, p.x_st, p.x_en, p.n_init = 1, 0, Lq - 1, 1
        p.obs_cap, p.factor_cap, p.z_cols, p.jitter = max(8, len(f["x"])), 4, 4, 0.0
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
        b.close()
        if return_std:
            return mean[0, :Lq].copy(), std[0, :Lq].copy()
        return mean[0, :Lq].copy()

    def sample_y(self, X, n_samples=1, random_state=0):
        raise NotImplementedError("posterior draws: GP_Edge_Tracing.fit_predict_GP (device eigen-factor sampler)")
