"""Converged final fit of ``fit_predict_GP`` (gpet.py:232-248, 262-266) -- SURVEY 8(f) row 2.

Host side for now: the reference runs scipy's L-BFGS-B (1 + 12 starts) over the log marginal
likelihood of the <= ~100-point training set (sklearn_gpr.py:254-295, 475-607).  The optimiser
is the reference's own third-party dependency and stays on the host; the objective is restated
here in NumPy and is scheduled to move into libgpet_hip.so ("what comes next").
This is NOT on the per-iteration hot path and is not a fallback for it.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg
import scipy.optimize


def _corr_and_grad(kernel_type, nu, xs, ell):
    a = xs / ell
    diff = a[:, None] - a[None, :]
    D = diff * diff
    if kernel_type == "RBF":
        R = np.exp(-0.5 * D)
        return R, R * D
    d = np.sqrt(D)
    if nu == 0.5:
        R = np.exp(-d)
        with np.errstate(divide="ignore", invalid="ignore"):
            G = np.where(d > 0, R * D / np.where(d > 0, d, 1.0), 0.0)
        return R, G
    if nu == 1.5:
        k = d * math.sqrt(3)
        return (1.0 + k) * np.exp(-k), 3 * D * np.exp(-np.sqrt(3 * D))
    if nu == 2.5:
        k = d * math.sqrt(5)
        tmp = np.sqrt(5 * D)
        return (1.0 + k + k ** 2 / 3.0) * np.exp(-k), 5.0 / 3.0 * D * (tmp + 1) * np.exp(-tmp)
    raise NotImplementedError("Matern nu must be 0.5, 1.5 or 2.5")


def _corr(kernel_type, nu, xa, xb, ell):
    diff = (xa / ell)[:, None] - (xb / ell)[None, :]
    D = diff * diff
    if kernel_type == "RBF":
        return np.exp(-0.5 * D)
    d = np.sqrt(D)
    if nu == 0.5:
        return np.exp(-d)
    if nu == 1.5:
        k = d * math.sqrt(3)
        return (1.0 + k) * np.exp(-k)
    k = d * math.sqrt(5)
    return (1.0 + k + k ** 2 / 3.0) * np.exp(-k)


def lml_and_grad(theta, xs, ys, w, kernel_type, nu, jitter=1e-6):
    """Log marginal likelihood and its gradient wrt theta = log(c, l, noise) (sklearn_gpr.py:512-585)."""
    c, ell, nl = np.exp(theta)
    R, dR = _corr_and_grad(kernel_type, nu, xs, ell)
    n = xs.shape[0]
    K = c * R + np.diag(nl * w)
    K[np.diag_indices(n)] += jitter
    try:
        L = scipy.linalg.cholesky(K, lower=True, check_finite=False)
    except np.linalg.LinAlgError:
        return -np.inf, np.zeros_like(theta)
    alpha = scipy.linalg.cho_solve((L, True), ys, check_finite=False)
    lml = -0.5 * ys @ alpha - np.log(np.diag(L)).sum() - n / 2 * np.log(2 * np.pi)
    Kinv = scipy.linalg.cho_solve((L, True), np.eye(n), check_finite=False)
    inner = np.outer(alpha, alpha) - Kinv
    g = np.array([0.5 * np.einsum("ij,ji->", inner, Gk) for Gk in (c * R, c * dR, np.diag(nl * w))])
    return lml, g


BOUNDS = np.log(np.array([[0.01, 1e3], [0.1, 100.0], [1e-18, 1.0]]))  # gpet.py:246-248


def prepare(init_sorted, obs_xy, x_grid, fix_endpoints):
    """Training set of the converged branch: sorted, standardised twice (gpet.py:235-238 then
    sklearn_gpr.py:229-234), with the transforms needed to map predictions back."""
    pts = np.concatenate([np.asarray(init_sorted).reshape(-1, 2), np.asarray(obs_xy).reshape(-1, 2)], axis=0)
    w = np.concatenate([np.full(len(init_sorted), 1e-7 if fix_endpoints else 0.5), np.ones(len(obs_xy))])
    order = np.argsort(pts[:, 0])
    pts, w = pts[order], w[order]
    x = pts[:, 0].astype(np.float64)
    y = pts[:, 1].astype(np.float64)
    y_m, y_s = np.mean(y), np.std(y)
    ys = (y - y_m) / y_s
    X_m, X_s = np.mean(x), np.std(x)
    xs = (x - X_m) / X_s
    xg = np.asarray(x_grid, dtype=np.float64)
    if xs.shape[0] == xg.shape[0]:  # sklearn_gpr.py:673-677
        w = np.zeros_like(w)
    m2, s2 = float(np.mean(ys)), float(np.std(ys))
    s2 = 1.0 if s2 == 0.0 else s2
    return dict(xs=xs, yt=(ys - m2) / s2, w=w, y_m=y_m, y_s=y_s, X_m=X_m, X_s=X_s, m2=m2, s2=s2, xg=xg)


def prepare_many(inits, obs_list, x_grids, fix_endpoints_list):
    """``prepare`` for many edges with far fewer NumPy calls: edges with the same numbers of init and observation
    points (most of a batch) are stacked and reduced along axis 1.  Row-wise ``argsort`` / ``mean`` / ``std`` on a
    C-contiguous 2-D array run the same per-row kernels as their 1-D forms, so every value is bit-identical to
    ``prepare`` (tests/test_host_logic.py)."""
    out = [None] * len(inits)
    groups = {}
    for e, (init, obs, xg, fe) in enumerate(zip(inits, obs_list, x_grids, fix_endpoints_list)):
        init = np.asarray(init).reshape(-1, 2)
        obs = np.asarray(obs).reshape(-1, 2)
        groups.setdefault((init.shape[0], obs.shape[0], len(xg), bool(fe), init.dtype.str, obs.dtype.str), []).append(e)
    for (ni, no, nxg, fe, _, _), members in groups.items():
        pts = np.stack([np.concatenate([np.asarray(inits[e]).reshape(-1, 2), np.asarray(obs_list[e]).reshape(-1, 2)], axis=0)
                        for e in members])                                          # (g, n, 2)
        w1 = np.concatenate([np.full(ni, 1e-7 if fe else 0.5), np.ones(no)])
        order = np.argsort(np.ascontiguousarray(pts[:, :, 0]), axis=1)
        pts = np.take_along_axis(pts, order[:, :, None], axis=1)
        w = w1[order]
        x = np.ascontiguousarray(pts[:, :, 0].astype(np.float64))
        y = np.ascontiguousarray(pts[:, :, 1].astype(np.float64))
        y_m, y_s = np.mean(y, axis=1), np.std(y, axis=1)
        ys = (y - y_m[:, None]) / y_s[:, None]
        X_m, X_s = np.mean(x, axis=1), np.std(x, axis=1)
        xs = (x - X_m[:, None]) / X_s[:, None]
        if ni + no == nxg:  # sklearn_gpr.py:673-677
            w = np.zeros_like(w)
        m2, s2 = np.mean(ys, axis=1), np.std(ys, axis=1)
        s2 = np.where(s2 == 0.0, 1.0, s2)
        yt = (ys - m2[:, None]) / s2[:, None]
        for k, e in enumerate(members):
            out[e] = dict(xs=xs[k], yt=yt[k], w=w[k], y_m=y_m[k], y_s=y_s[k], X_m=X_m[k], X_s=X_s[k],
                          m2=float(m2[k]), s2=float(s2[k]), xg=np.asarray(x_grids[e], dtype=np.float64))
    return out


def start_points_many(noise_ys, seeds, n_restarts=12):
    """``start_points`` for many edges.  ``RandomState(seed)`` costs ~50 us of Python per edge; here the generator is
    evaluated for all seeds at once: MT19937's ``init_genrand`` recurrence (what numpy's legacy seeding runs for an integer
    seed), the first twist of the state, the tempering and ``random_sample``'s 53-bit doubles
    ``((a >> 5) * 2**26 + (b >> 6)) / 2**53`` -- ``uniform(0, 1)`` is ``0.0 + 1.0 * random_sample()``, the same value.
    Bit-identical to ``start_points`` (tests/test_host_logic.py)."""
    seeds = np.asarray(seeds, dtype=np.uint64)
    if np.any(seeds > np.uint64(0xFFFFFFFF)):
        raise ValueError("seeds must fit 32 bits (numpy legacy seeding)")
    E = seeds.shape[0]
    n_u = 3 * max(0, n_restarts)          # doubles needed per edge
    n_w = 2 * n_u                         # 32-bit outputs needed per edge
    th0 = np.empty((E, 1 + max(0, n_restarts), 3))
    th0[:, 0, 0] = th0[:, 0, 1] = np.log(5.0)
    th0[:, 0, 2] = np.log(np.asarray(noise_ys, dtype=np.float64))
    if n_restarts > 0:
        if n_w > 227:
            raise ValueError("start_points_many: more than 37 restarts need a second twist phase")
        n_key = n_w + 397 + 1             # state words the first n_w outputs depend on
        key = np.empty((n_key, E), dtype=np.uint64)
        key[0] = seeds
        mask = np.uint64(0xFFFFFFFF)
        for i in range(1, n_key):
            prev = key[i - 1]
            key[i] = (np.uint64(1812433253) * (prev ^ (prev >> np.uint64(30))) + np.uint64(i)) & mask
        # genrand: word k of the next state = key[k + 397] ^ twist(key[k], key[k + 1]) for k < 227
        y = (key[:n_w] & np.uint64(0x80000000)) | (key[1:n_w + 1] & np.uint64(0x7FFFFFFF))
        w = key[397:397 + n_w] ^ (y >> np.uint64(1)) ^ np.where((y & np.uint64(1)) != 0, np.uint64(0x9908B0DF), np.uint64(0))
        w ^= w >> np.uint64(11)
        w ^= (w << np.uint64(7)) & np.uint64(0x9D2C5680)
        w ^= (w << np.uint64(15)) & np.uint64(0xEFC60000)
        w &= mask
        w ^= w >> np.uint64(18)
        a = (w[0::2] >> np.uint64(5)).astype(np.float64)
        b_ = (w[1::2] >> np.uint64(6)).astype(np.float64)
        u = ((a * 67108864.0 + b_) / 9007199254740992.0).T.reshape(E, n_restarts, 3)  # row-major fill of uniform(size=(R, 3))
        lo, span = BOUNDS[:, 0], BOUNDS[:, 1] - BOUNDS[:, 0]
        th0[:, 1:, :] = lo + span * u
    return list(th0.reshape(-1, 3))


def start_points(noise_y, seed, n_restarts=12):
    """theta of the kernel (gpet.py:244-245) + log-uniform restarts (sklearn_gpr.py:283-288)."""
    th = np.empty((1 + max(0, n_restarts), 3))
    th[0] = np.log(np.array([5.0, 5.0, float(noise_y)]))
    if n_restarts > 0:
        u = np.random.RandomState(seed).uniform(size=(n_restarts, 3))
        th[1:] = BOUNDS[:, 0] + (BOUNDS[:, 1] - BOUNDS[:, 0]) * u  # same elementwise arithmetic as row by row
    return list(th)


def finish(prep, theta, kernel_type, nu):
    """Posterior mean (pixels) and std (standardised units, gpet.py:266) at the optimum."""
    c, ell, nl = np.exp(theta)
    xs, yt, w = prep["xs"], prep["yt"], prep["w"]
    n = xs.shape[0]
    K = c * _corr(kernel_type, nu, xs, xs, ell) + np.diag(nl * w)
    K[np.diag_indices(n)] += 1e-6
    L = scipy.linalg.cholesky(K, lower=True, check_finite=False)
    alpha = scipy.linalg.cho_solve((L, True), yt, check_finite=False)
    xq = (prep["xg"] - prep["X_m"]) / prep["X_s"]
    Kt = c * _corr(kernel_type, nu, xq, xs, ell)
    mean = prep["s2"] * (Kt @ alpha) + prep["m2"]
    V = scipy.linalg.solve_triangular(L, Kt.T, lower=True, check_finite=False)
    var = np.full(xq.shape[0], c) - np.einsum("ij,ji->i", V.T, V)
    var[var < 0] = 0.0
    return prep["y_s"] * mean + prep["y_m"], np.sqrt(var * prep["s2"] ** 2)


def converged_fit_predict(init_sorted, obs_xy, x_grid, kernel_type, nu, noise_y, fix_endpoints, seed,
                          n_restarts=12):
    """Returns (y_mean in pixels, y_std in standardised units -- the reference does not rescale
    it, gpet.py:266 --, theta)."""
    pts = np.concatenate([np.asarray(init_sorted).reshape(-1, 2), np.asarray(obs_xy).reshape(-1, 2)], axis=0)
    w = np.concatenate([np.full(len(init_sorted), 1e-7 if fix_endpoints else 0.5), np.ones(len(obs_xy))])
    order = np.argsort(pts[:, 0])
    pts, w = pts[order], w[order]
    x = pts[:, 0].astype(np.float64)
    y = pts[:, 1].astype(np.float64)
    y_m, y_s = np.mean(y), np.std(y)
    ys = (y - y_m) / y_s
    X_m, X_s = np.mean(x), np.std(x)
    xs = (x - X_m) / X_s
    xg = np.asarray(x_grid, dtype=np.float64)
    if xs.shape[0] == xg.shape[0]:  # sklearn_gpr.py:673-677
        w = np.zeros_like(w)
    m2, s2 = float(np.mean(ys)), float(np.std(ys))  # normalize_y=False standardises (sklearn_gpr.py:229-234)
    s2 = 1.0 if s2 == 0.0 else s2
    yt = (ys - m2) / s2
    bounds = np.log(np.array([[0.01, 1e3], [0.1, 100.0], [1e-18, 1.0]]))  # gpet.py:246-248
    theta0 = np.log(np.array([5.0, 5.0, float(noise_y)]))  # gpet.py:244-245

    def obj(th):
        lml, g = lml_and_grad(th, xs, yt, w, kernel_type, nu)
        return -lml, -g

    def run(th0):
        r = scipy.optimize.minimize(obj, th0, method="L-BFGS-B", jac=True, bounds=bounds)
        return r.x, r.fun

    optima = [run(theta0)]
    if n_restarts > 0:
        u = np.random.RandomState(seed).uniform(size=(n_restarts, 3))  # sklearn_gpr.py:205,285
        for r in range(n_restarts):
            optima.append(run(bounds[:, 0] + (bounds[:, 1] - bounds[:, 0]) * u[r]))
    theta = optima[int(np.argmin([o[1] for o in optima]))][0]
    c, ell, nl = np.exp(theta)
    n = xs.shape[0]
    K = c * _corr(kernel_type, nu, xs, xs, ell) + np.diag(nl * w)
    K[np.diag_indices(n)] += 1e-6
    L = scipy.linalg.cholesky(K, lower=True, check_finite=False)
    alpha = scipy.linalg.cho_solve((L, True), yt, check_finite=False)
    xq = (xg - X_m) / X_s
    Kt = c * _corr(kernel_type, nu, xq, xs, ell)
    mean = s2 * (Kt @ alpha) + m2
    V = scipy.linalg.solve_triangular(L, Kt.T, lower=True, check_finite=False)
    var = np.full(xq.shape[0], c) - np.einsum("ij,ji->i", V.T, V)
    var[var < 0] = 0.0
    return y_s * mean + y_m, np.sqrt(var * s2 ** 2), theta
