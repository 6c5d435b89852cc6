"""Would a warm start pay for the ANY-RANK factor (one-sided Jacobi on the rows of a factor G, G^T G = Sigma; csrc/gpet_eig.hip)?
Matern-5/2 traces on an N-column image: posterior covariance of every iteration, rows orthogonalised by round-robin
one-sided Jacobi (relative stopping test 1e-8 per pair, as the device) starting
  cold: from the pivoted Cholesky factor of Sigma (what the device does), and
  warm: from X = C V^T with V the PREVIOUS iteration's eigenvectors and C^T C = V^T Sigma V (a plain Cholesky factor of a
        nearly diagonal matrix): X^T X = Sigma and the rows of X are nearly orthogonal already.
CPU only:  python tests/analysis/onesided_warm_start.py [N]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import scipy.linalg  # noqa: E402
from oracle import gpet_oracle as orc  # noqa: E402


def rr_pairs(n, rnd):
    """round-robin pairing of n (even) players in round rnd"""
    idx = np.arange(n)
    a = np.concatenate([[n - 1], (idx[:n - 1] + rnd) % (n - 1)])
    return a[:n // 2], a[n - 1:n // 2 - 1:-1]


def onesided_sweeps(G, tol=1e-8, max_sweeps=40):
    G = G.copy()
    n = G.shape[0]
    if n % 2:
        G = np.vstack([G, np.zeros((1, G.shape[1]))])
        n += 1
    for sweep in range(max_sweeps):
        worst = 0.0
        for rnd in range(n - 1):
            p, q = rr_pairs(n, rnd)
            gp, gq = G[p], G[q]
            app = np.einsum("ij,ij->i", gp, gp)
            aqq = np.einsum("ij,ij->i", gq, gq)
            apq = np.einsum("ij,ij->i", gp, gq)
            den = np.sqrt(app * aqq)
            rel = np.where(den > 0, np.abs(apq) / np.where(den > 0, den, 1.0), 0.0)
            worst = max(worst, float(rel.max()))
            act = rel > 1e-18
            d = aqq - app
            t = np.where(act, np.sign(np.where(d == 0, 1.0, d)) * 2.0 * apq / (np.abs(d) + np.sqrt(d * d + 4.0 * apq * apq) + 1e-300), 0.0)
            c = 1.0 / np.sqrt(1.0 + t * t)
            s = t * c
            G[p] = c[:, None] * gp - s[:, None] * gq
            G[q] = s[:, None] * gp + c[:, None] * gq
        if worst <= tol:
            return sweep + 1
    return max_sweeps


def pivoted_cholesky(S, tol=1e-14):
    n = S.shape[0]
    d = np.diag(S).copy()
    G = np.zeros((0, n))
    dmax = d.max()
    while G.shape[0] < n:
        j = int(np.argmax(d))
        if d[j] <= tol * dmax:
            break
        row = (S[j] - G[:, j] @ G) / np.sqrt(d[j])
        G = np.vstack([G, row])
        d = d - row * row
        d[j] = -np.inf
    return G


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    kw = dict(kernel_options={'kernel': 'Matern', 'sigma_f': 154.0 * N / 1024, 'length_scale': 41.0 * N / 1024, 'nu': 2.5}, noise_y=1, N_samples=200,
              score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    img, edge = orc.synth_sinusoid_image(N, 5)
    grad = orc.comp_grad_img(img, orc.kernel_builder((11, 5)))
    init = edge[[0, -1], :][:, [1, 0]]
    rec = []
    orc.trace(init, grad, seed=3, record=rec, sign_convention="harmonic", **kw)
    p = orc.resolve_params(init, grad, seed=3, **kw)
    xg = np.asarray(p["x_grid"], dtype=np.float64)
    Vprev = None
    for t, r in enumerate(rec):
        _, inf = orc.fit_predict_samples(p["init"], r["obs_in"], p, 1, want_all=True, Z=np.zeros((2, len(xg))))
        cov = inf["pred"]["cov"]
        cov = 0.5 * (cov + cov.T)
        G = pivoted_cholesky(cov)
        cold = onesided_sweeps(G)
        warm = None
        if Vprev is not None:
            M = Vprev.T @ cov @ Vprev
            M = 0.5 * (M + M.T)
            try:
                C = scipy.linalg.cholesky(M, lower=False)
                warm = onesided_sweeps(C @ Vprev.T)
            except np.linalg.LinAlgError:
                warm = "Cholesky of V^T Sigma V failed"
        s, V = np.linalg.eigh(cov)
        print("iteration %2d: %3d observations, rank of the pivoted Cholesky %d of %d: sweeps cold %d, warm %s" % (t, len(r["obs_in"]), G.shape[0], len(xg), cold, warm), flush=True)
        Vprev = V[:, ::-1]


if __name__ == "__main__":
    main()
