"""Host-side helpers around the converged fit of ``fit_predict_GP`` (gpet.py:232-248, 262-266).

The converged fit itself runs on the device (``gpet_final_fit_all``: standardisation, start points, L-BFGS-B, best
restart, posterior -- csrc/gpet_lbfgsb.hip).  What is left here restates, in NumPy, only the INPUTS of the per-stage
objective entry points (``gpet_final_set_training`` + ``gpet_lml_batch``) for callers who drive an optimiser of their
own, and is what the tests compare the device's training sets and start points against: nothing in this module
evaluates a likelihood.
"""
from __future__ import annotations

import numpy as np

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


def start_points(noise_y, seed, n_restarts=12):
    """theta of the kernel (gpet.py:244-245) + log-uniform restarts (sklearn_gpr.py:283-288)."""
    th = np.empty((1 + max(0, n_restarts), 3))
    th[0] = np.log(np.array([5.0, 5.0, float(noise_y)]))
    if n_restarts > 0:
        u = np.random.RandomState(seed).uniform(size=(n_restarts, 3))
        th[1:] = BOUNDS[:, 0] + (BOUNDS[:, 1] - BOUNDS[:, 0]) * u  # same elementwise arithmetic as row by row
    return list(th)
