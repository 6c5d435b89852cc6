"""Would a warm start pay for the per-iteration eigen-decomposition of the structured path (k_jacobi_seat)?
H_t = Q0^T Sigma_t Q0 (Q0: the r0 leading eigenvectors of the prior correlation on the grid) of every iteration of a README
trace, diagonalised by cyclic Jacobi sweeps to the kernel's stopping test (off^2 <= 1e-24 diag^2) from the identity and
from the previous iteration's eigenvectors.  CPU only (the oracle's GP):  python tests/analysis/jacobi_warm_start.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from oracle import gpet_oracle as orc  # noqa: E402

KW = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, noise_y=1, N_samples=1000, score_thresh=1,
          delta_x=5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)


def sweeps_needed(H, W0=None, max_sweeps=40):
    A = H.copy() if W0 is None else W0.T @ H @ W0
    n = A.shape[0]
    for sweep in range(max_sweeps):
        d = np.diag(A)
        off = np.sum((A - np.diag(d)) ** 2)  # (summed directly, as the kernel does: no cancellation)
        if off <= 1e-24 * np.sum(d * d):
            return sweep
        for p in range(n - 1):
            for q in range(p + 1, n):
                apq = A[p, q]
                if abs(apq) <= 1e-300 or apq * apq <= 1e-36 * abs(A[p, p] * A[q, q]):
                    continue
                dd = A[q, q] - A[p, p]
                t = (1.0 if dd >= 0 else -1.0) * 2.0 * apq / (abs(dd) + np.hypot(dd, 2.0 * apq))
                c = 1.0 / np.sqrt(1.0 + t * t)
                s = t * c
                rp, rq = A[p].copy(), A[q].copy()
                A[p], A[q] = c * rp - s * rq, s * rp + c * rq
                cp, cq = A[:, p].copy(), A[:, q].copy()
                A[:, p], A[:, q] = c * cp - s * cq, s * cp + c * cq
    return max_sweeps


def main():
    img, edge = orc.synth_sinusoid_image(500, 3)
    grad = orc.comp_grad_img(img, orc.kernel_builder((11, 5)))
    init = edge[[0, -1], :][:, [1, 0]]
    rec = []
    orc.trace(init, grad, seed=1, record=rec, sign_convention="harmonic", **KW)
    p = orc.resolve_params(init, grad, seed=1, **KW)
    xg = np.asarray(p["x_grid"], dtype=np.float64)
    rho = orc.corr_matrix(p["kernel_type"], p["nu"], xg, xg, p["length_scale"])
    lam, Q = np.linalg.eigh(rho)
    keep = lam > 1e-14 * lam.max()
    Q0 = Q[:, keep][:, ::-1]
    lam0 = lam[keep][::-1]
    print("prior rank r0 = %d" % Q0.shape[1])
    Wprev = None
    for t, r in enumerate(rec):
        # the structured path's own formula: Sigma / y_std^2 = Q0 (c Lam - U^T U) Q0^T, U = L^-1 (c Q0[obs] Lam)
        x, y, w = orc.assemble_training(p["init"], r["obs_in"], p["fix_endpoints"])
        y_s = float(np.std(y)) + 1.0
        amp = p["sigma_f"] ** 2 / y_s ** 2
        fit = orc.gp_fit(x, y / y_s, w, amp, p["length_scale"], p["kernel_type"], p["nu"], p["noise_y"], len(xg))
        idx = np.rint(x - xg[0]).astype(int)
        import scipy.linalg
        U = scipy.linalg.solve_triangular(fit["L"], amp * Q0[idx, :] * lam0[None, :], lower=True, check_finite=False)
        H = amp * np.diag(lam0) - U.T @ U
        H = 0.5 * (H + H.T)
        cold = sweeps_needed(H)
        warm = sweeps_needed(H, Wprev) if Wprev is not None else None
        th, W = np.linalg.eigh(H)
        print("iteration %2d: %3d observations, sweeps from the identity %d, from the previous eigenvectors %s" % (t, len(r["obs_in"]), cold, warm))
        Wprev = W


if __name__ == "__main__":
    main()
