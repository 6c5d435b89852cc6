"""Many independent L-BFGS-B minimisations advanced in lock step.

``scipy.optimize.minimize(method="L-BFGS-B")`` (what the reference's GaussianProcessRegressor
uses, sklearn_gpr.py:587-595) is a reverse-communication loop around ``_lbfgsb.setulb``
(scipy/optimize/_lbfgsb_py.py).  Driving that same routine for all (edge, restart) problems at
once lets every round of objective evaluations run as ONE batched GPU launch, while each
problem follows exactly the iterates scipy's own driver would produce for the same f/g values.
Falls back to plain ``scipy.optimize.minimize`` per problem if the private routine is missing.
"""
from __future__ import annotations

import numpy as np

try:  # private but stable across the scipy versions this image ships
    from scipy.optimize import _lbfgsb
    _HAVE_SETULB = hasattr(_lbfgsb, "setulb")
except Exception:  # pragma: no cover
    _lbfgsb = None
    _HAVE_SETULB = False


class _Problem:
    __slots__ = ("x", "f", "g", "wa", "iwa", "task", "ln_task", "lsave", "isave", "dsave", "n_iter", "nfev",
                 "state", "low", "up", "nbd")

    def __init__(self, x0, bounds, m):
        n = x0.shape[0]
        lo, hi = bounds[:, 0], bounds[:, 1]
        self.x = np.clip(np.array(x0, dtype=np.float64), lo, hi)
        self.f = 0.0
        self.g = np.zeros(n, dtype=np.float64)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32)
        self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32)
        self.isave = np.zeros(44, dtype=np.int32)
        self.dsave = np.zeros(29, dtype=np.float64)
        self.low = np.where(np.isinf(lo), 0.0, lo).astype(np.float64)
        self.up = np.where(np.isinf(hi), 0.0, hi).astype(np.float64)
        code = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}
        self.nbd = np.array([code[(bool(np.isfinite(a)), bool(np.isfinite(b)))] for a, b in zip(lo, hi)],
                            dtype=np.int32)
        self.n_iter = 0
        self.nfev = 0
        self.state = "run"  # run | want_fg | done


def minimize_many(eval_batch, x0s, bounds, m=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxiter=15000,
                  maxfun=15000, maxls=20):
    """Minimise len(x0s) problems that share ``bounds`` (n, 2).

    ``eval_batch(idx, X)``: idx = list of problem indices, X = (len(idx), n) points;
    returns (f (len(idx),), g (len(idx), n)).  Returns (xs (P, n), funs (P,), n_rounds)."""
    bounds = np.asarray(bounds, dtype=np.float64)
    P = len(x0s)
    if not _HAVE_SETULB:  # pragma: no cover - same results, one problem at a time
        import scipy.optimize
        xs, fs = [], []
        for i, x0 in enumerate(x0s):
            def fun(x, i=i):
                f, g = eval_batch([i], x[None, :])
                return float(f[0]), g[0]
            r = scipy.optimize.minimize(fun, x0, method="L-BFGS-B", jac=True, bounds=bounds)
            xs.append(r.x)
            fs.append(r.fun)
        return np.array(xs), np.array(fs), -1
    factr = ftol / np.finfo(float).eps
    probs = [_Problem(np.asarray(x0, dtype=np.float64), bounds, m) for x0 in x0s]
    rounds = 0
    while True:
        pending = []
        for i, p in enumerate(probs):
            while p.state == "run":
                _lbfgsb.setulb(m, p.x, p.low, p.up, p.nbd, p.f, p.g, factr, gtol, p.wa, p.iwa, p.task, p.lsave,
                               p.isave, p.dsave, maxls, p.ln_task)
                if p.task[0] == 3:
                    p.state = "want_fg"
                elif p.task[0] == 1:
                    p.n_iter += 1
                    if p.n_iter >= maxiter:
                        p.task[0], p.task[1] = 5, 504
                    elif p.nfev > maxfun:
                        p.task[0], p.task[1] = 5, 502
                else:
                    p.state = "done"
            if p.state == "want_fg":
                pending.append(i)
        if not pending:
            break
        X = np.stack([probs[i].x for i in pending])
        f, g = eval_batch(pending, X)
        for k, i in enumerate(pending):
            p = probs[i]
            p.f = float(f[k])
            p.g = np.array(g[k], dtype=np.float64)
            p.nfev += 1
            p.state = "run"
        rounds += 1
    return np.stack([p.x for p in probs]), np.array([p.f for p in probs]), rounds
