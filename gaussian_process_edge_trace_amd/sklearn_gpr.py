"""``GaussianProcessRegressor`` / ``WeightedWhiteKernel`` mirrors of ``gp_edge_tracing/sklearn_gpr.py``.

The reference vendors a modified scikit-learn GPR (sklearn_gpr.py:31-610) and uses it only inside
``GP_Edge_Tracing.fit_predict_GP``.  This mirror keeps the parts of that interface the edge tracer
exercises -- ``fit`` with ``optimizer=None`` and ``predict`` (mean / std) -- with the reference's
modified ``normalize_y`` semantics (True centres only, False standardises; sklearn_gpr.py:221-234)
and runs them in libgpet_hip.so (the kernels of the converged fit: Cholesky + alpha + forward
substitution per query point).  Query points must be an arithmetic progression (the tracer's
x-grid): the device kernels index the posterior by grid position.  ``predict(return_cov=True)``, ``sample_y`` (the
device eigen-factor sampler on the query grid) and ``log_marginal_likelihood`` run on the device too, and so does
``optimizer="fmin_l_bfgs_b"`` (sklearn_gpr.py:254-295, 587-607): theta0 plus ``n_restarts_optimizer`` log-uniform
restarts drawn like the reference draws them, minimised by the device L-BFGS-B of the tracer's converged fit
(``gpet_final_optimize``) inside the kernel's bounds.  The device objective carries its regulariser as
``noise_level * w_i + 1e-6`` on the diagonal, so the optimiser needs ``alpha == 1e-6`` when a ``WeightedWhiteKernel`` is
present (the reference's only use, gpet.py:154-158); without one any ``alpha`` is folded into fixed weights.  A callable
optimiser is not supported.
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


def _log_bounds(obj, name, value):
    """log bounds of one hyper-parameter as scikit-learn kernels carry them (``<name>_bounds``; "fixed" -> lo == hi)."""
    b = getattr(obj, name + "_bounds", None) if obj is not None else None
    if b is None or isinstance(b, str):
        return [np.log(value), np.log(value)] if isinstance(b, str) or obj is None else [np.log(1e-5), np.log(1e5)]
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return [float(np.log(b[0])), float(np.log(b[1]))]


def _kernel_bounds(kernel, const, ell, white):
    """(3, 2) log bounds of (constant, length_scale, noise_level); a dict kernel may carry 'bounds' = that array in linear units."""
    if isinstance(kernel, dict):
        b = kernel.get("bounds")
        if b is not None:
            return np.log(np.asarray(b, dtype=np.float64).reshape(3, 2))
        lo_hi = [np.log(1e-5), np.log(1e5)]
        return np.array([lo_hi, lo_hi, lo_hi if white is not None else [0.0, 0.0]])
    k = kernel.k1 if hasattr(kernel, "k1") and isinstance(getattr(kernel, "k2", None), WeightedWhiteKernel) else kernel
    rows = [_log_bounds(k.k1, "constant_value", const), _log_bounds(k.k2, "length_scale", ell)]
    rows.append(_log_bounds(white, "noise_level", white.noise_level) if white is not None else [0.0, 0.0])
    return np.array(rows)


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
    _CACHE_MAX = 2  # device batches kept per regressor (one for predict/LML, one for sample_y is the usual pair)

    def __init__(self, kernel=None, alpha=1e-10, optimizer=None, n_restarts_optimizer=0, normalize_y=False,
                 copy_X_train=True, random_state=None, *, device=0, _ctx=None):
        if optimizer is not None and optimizer != "fmin_l_bfgs_b":
            raise NotImplementedError("optimizer must be None or 'fmin_l_bfgs_b' (the device L-BFGS-B); callables are not supported")
        self.kernel = kernel
        self._batches = {}  # device batches of _device_fit_predict, by (grid length, samples, factor): built once, reused
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
        if n > 16384:
            raise NotImplementedError("more than 16384 training points (the n x n kernel matrix lives in HBM per edge)")
        self._fit = dict(kt=kt, nu=nu, const=const, ell=ell, x=X[:, 0].copy(), yt=yt, noise=nl * w + alpha)
        self.X_train_, self.y_train_ = X, yt
        self._lml_value = None  # optimizer=None: evaluated on first access (sklearn_gpr.py:296-299 computes it in fit)
        if self.optimizer is not None:
            self._optimise(const, ell, nl, w, alpha, white)
        return self

    def _optimise(self, const, ell, nl, w, alpha, white):
        """sklearn_gpr.py:254-295: maximise the log marginal likelihood from theta0 and n_restarts_optimizer log-uniform
        restarts (RandomState(random_state).uniform over the log bounds, one restart at a time like the reference), keep
        the best; on the device (gpet_final_optimize)."""
        f = self._fit
        bounds = _kernel_bounds(self.kernel, const, ell, white)
        if white is not None:
            if not np.allclose(alpha, 1e-6, rtol=0, atol=1e-18):
                raise NotImplementedError("optimizer with a WeightedWhiteKernel needs alpha == 1e-6 (the objective's fixed jitter)")
            weights, theta0 = w, np.log([const, ell, nl])
        else:
            weights, theta0 = alpha - 1e-6, np.log([const, ell, 1.0])  # K_ii = c + 1 * (alpha - 1e-6) + 1e-6
            bounds[2] = [0.0, 0.0]
        starts = [theta0]
        if self.n_restarts_optimizer > 0:
            if not np.all(np.isfinite(bounds)):
                raise ValueError("Multiple optimizer restarts (n_restarts_optimizer>0) requires that all bounds are finite.")
            rng = self.random_state if isinstance(self.random_state, np.random.RandomState) else np.random.RandomState(self.random_state)
            # the reference draws over kernel_.bounds (sklearn_gpr.py:283-288): the NON-FIXED hyper-parameters only, in theta
            # order -- a "fixed" one (lo == hi) or the absent noise level takes no number from the stream
            free = bounds[:, 0] != bounds[:, 1]
            for _ in range(self.n_restarts_optimizer):
                th = np.array(theta0, dtype=np.float64)
                th[free] = rng.uniform(bounds[free, 0], bounds[free, 1])
                starts.append(th)
        b = self._batch_for(np.arange(4.0), 1, False)
        b.final_set_training_all([f["x"]], [f["yt"]], [weights])
        theta, fmin, self._opt_rounds = b.final_optimize(np.asarray(starts)[None], bounds)
        c_opt, l_opt, nl_opt = np.exp(theta[0])
        self._lml_value = -float(fmin[0])
        self.kernel_theta_ = theta[0].copy()
        f.update(const=float(c_opt), ell=float(l_opt), noise=(nl_opt * w + alpha) if white is not None else f["noise"])
        self._fit_nl = float(nl_opt) if white is not None else 1.0

    @property
    def log_marginal_likelihood_value_(self):
        """Log marginal likelihood at the fitted hyper-parameters: the optimum's with an optimiser; with
        ``optimizer=None`` the reference evaluates it inside ``fit`` (sklearn_gpr.py:296-299) -- here on first access,
        so the tracer's fits do not pay a second Cholesky for a number nobody reads."""
        if self._lml_value is None and getattr(self, "_fit", None) is not None:
            self._lml_value = float(self.log_marginal_likelihood())
        return self._lml_value

    def _batch_for(self, X, n_samples, want_factor):
        """The device batch of one edge whose grid is the query progression: built once per (length, samples, factor)
        and kept (a 4 x Lq image upload, its gradient KDE and the arena cost more than the fit)."""
        f = self._fit
        xq = np.asarray(X, dtype=np.float64).reshape(-1)
        Lq = xq.shape[0]
        # training capacity rounded up to a power of two: refits with a few more points reuse the arena.  The batch's own
        # length scale is NOT part of the key: every call passes the hyper-parameters through `par`
        # (gpet_final_predict_all / gpet_lml_batch), params.length_scale only sizes the construction-time defaults.
        cap = 8
        while cap < len(f["x"]):
            cap *= 2
        key = (Lq, max(1, int(n_samples)), bool(want_factor), f["kt"], f["nu"], cap)
        b = self._batches.pop(key, None)
        if b is not None:
            self._batches[key] = b  # (most recently used last)
        else:
            while len(self._batches) >= self._CACHE_MAX:  # an arena holds an n_cap x n_cap K: evict the least recently used
                self._batches.pop(next(iter(self._batches))).close()
            p = _lib.GpetParams()
            p.kernel_type = _lib.KERNEL_MATERN if f["kt"] == "Matern" else _lib.KERNEL_RBF
            p.nu, p.sigma_f, p.length_scale, p.noise_y = f["nu"], 1.0, f["ell"], 1.0
            p.n_samples, p.n_keep, p.delta_x, p.pixel_thresh, p.score_thresh = key[1], 1, max(2, Lq // 4), 2, 1.0
            p.fix_endpoints, p.x_st, p.x_en, p.n_init = 1, 0, Lq - 1, 1
            p.obs_cap, p.jitter = key[5], 0.0
            p.factor_cap, p.z_cols = (Lq, Lq) if want_factor else (4, 4)
            img = np.zeros((4, Lq), dtype=np.float32)
            img[0, 0] = 1.0
            b = _lib.Batch(self._ctx, [img], [p], [np.array([[0, 0]], dtype=np.int64)])
            self._batches[key] = b
        return b

    def close(self):
        for b in self._batches.values():
            b.close()
        self._batches = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _device_fit_predict(self, X, n_samples=1, want_factor=False):
        """A (cached) batch of a single edge whose grid is 0..Lq-1 ((x - X_m)/X_s maps it onto the query points):
        the converged-fit kernels evaluate K, its Cholesky factor, alpha and the posterior at the queries."""
        f = self._fit
        xq = np.asarray(X, dtype=np.float64).reshape(-1)
        Lq = xq.shape[0]
        step = (xq[-1] - xq[0]) / (Lq - 1) if Lq > 1 else 1.0
        if Lq < 4 or step <= 0 or not np.allclose(xq, xq[0] + step * np.arange(Lq), rtol=0, atol=1e-9 * max(1.0, abs(step))):
            raise NotImplementedError("query points must be an increasing arithmetic progression of >= 4 points")
        b = self._batch_for(X, n_samples, want_factor)
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
        if return_cov:
            b.final_cov()  # (K** - V^T V) * y_train_std^2 on the f64 matrix cores
            return mean, b.read(_lib.BUF_COV)
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
        b.final_cov()
        b.factor()
        b.normals([int(random_state)])
        sc = b.scalars()
        sc.y_s = 1.0
        b.write_scalars(sc)
        b.write(_lib.BUF_MEAN, mean)
        b.sample()
        return b.read(_lib.BUF_SAMPLES).T.copy()

    def log_marginal_likelihood(self, theta=None, eval_gradient=False):
        """Log marginal likelihood of theta = log(constant, length_scale, noise_level) on the training set of ``fit``
        [and its gradient] (sklearn_gpr.py:475-585), evaluated by the batched objective kernel of the converged fits;
        theta=None: at the kernel's own hyper-parameters."""
        f = self._fit
        if theta is None:
            theta = np.log([f["const"], f["ell"], 1.0])
        b, _, _ = self._device_fit_predict(np.arange(4.0))
        val, grad = b.lml_batch([0], np.asarray(theta, dtype=np.float64).reshape(1, 3))
        if eval_gradient:
            return -float(val[0]), -grad[0]
        return -float(val[0])
