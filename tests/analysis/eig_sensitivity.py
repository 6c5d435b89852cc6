"""Analysis script (not a test, not product): how sensitive is a full-rank (Matern) trace to the
eigen-solver that factors the posterior covariance?  Runs the CPU oracle's trace with alternative
factorisations of the SAME covariance and reports the first iteration whose observation set differs
from the LAPACK-SVD run, plus the largest sample difference per iteration.

usage: python tests/analysis/eig_sensitivity.py [N] [cold]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gpet_oracle as orc  # noqa: E402


def harmonic(F):
    dots = F @ (1.0 / np.arange(1, F.shape[1] + 1))
    return np.where(dots < 0, -1.0, 1.0)[:, None] * F


def f_svd(cov, sign_convention=None):
    _, s, v = np.linalg.svd(cov)
    return harmonic(np.sqrt(s)[:, None] * v), s, v


def f_eigh(cov, sign_convention=None):
    w, V = np.linalg.eigh(cov)
    o = np.argsort(-np.abs(w), kind="stable")
    s = np.abs(w[o])
    v = V[:, o].T
    return harmonic(np.sqrt(s)[:, None] * v), s, v


def make_noise(rel):
    def f(cov, sign_convention=None):
        rng = np.random.default_rng(0)
        E = rng.standard_normal(cov.shape)
        E = (E + E.T) * (0.5 * rel * np.abs(cov).max())
        return f_svd(cov + E)
    return f


def f_chol_jacobi(cov, sign_convention=None):
    """eigh of L^T L (what a Cholesky-preconditioned one-sided Jacobi diagonalises), mapped back: rows = W^T L^T."""
    L = np.linalg.cholesky(cov)
    w, W = np.linalg.eigh(L.T @ L)
    o = np.argsort(-w, kind="stable")
    F = (L @ W[:, o]).T
    s = np.sum(F * F, axis=1)
    return harmonic(F), s, F / np.sqrt(s)[:, None]


def run(N, which, cold=False):
    img, truth = orc.synth_sinusoid_image(N, 5)
    grad = orc.comp_grad_img(img, orc.kernel_builder((11, 5)))
    init = truth[[0, -1], :][:, [1, 0]]
    warm = truth[16:-16:16][:, [1, 0]].astype(np.int64)
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N},
              noise_y=1, N_samples=300, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, seed=3,
              fix_endpoints=True)
    orc.mvn_factor_svd = which
    rec = []
    et, ci, info = orc.trace(init, grad, obs=(np.zeros((0, 2), dtype=np.int64) if cold else warm), record=rec, sign_convention="harmonic", **kw)
    return et, rec


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    solvers = dict(svd=f_svd, eigh=f_eigh, noise1e16=make_noise(1e-16), noise1e14=make_noise(1e-14),
                   noise1e12=make_noise(1e-12), chol=f_chol_jacobi)
    base = None
    for name, f in solvers.items():
        t = time.time()
        try:
            et, rec = run(N, f, cold=len(sys.argv) > 2 and sys.argv[2] == 'cold')
        except Exception as e:  # e.g. Cholesky of a semidefinite covariance
            print(name, "failed:", repr(e))
            continue
        if base is None:
            base = (et, rec)
            print("%-10s %d iterations (%.1f s)" % (name, len(rec), time.time() - t))
            continue
        first = next((i for i, (a, b) in enumerate(zip(rec, base[1])) if not np.array_equal(a["obs_out"], b["obs_out"])), None)
        idx = next((i for i, (a, b) in enumerate(zip(rec, base[1])) if not np.array_equal(a["best_idxs"], b["best_idxs"])), None)
        print("%-10s %d iterations, first differing obs set: %s, first differing best_idxs: %s, trace max diff %d px (%.1f s)"
              % (name, len(rec), first, idx, int(np.abs(et[:, 0] - base[0][:, 0]).max()) if et.shape == base[0].shape else -1,
                 time.time() - t))
