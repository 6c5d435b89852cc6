"""GPU parity at the other BASELINE.json configurations' shapes (they are parity cases, not
bench lines): config 3 (2048x2048, ~1500 observations, N_samples=4000: large-n Cholesky / TRSM
path) and config 5's frame shape (1024x1024, Matern-5/2, warm start)."""
import numpy as np
import pytest

from oracle import gpet_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import gaussian_process_edge_trace_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def ctx(amd):
    return amd._lib.Context(0)


def test_config3_large_n_gp_iteration(amd, ctx):
    """2048^2 image, 1498 user-supplied observations (+2 inits = 1500 training points), S=4000:
    one GP iteration + scoring.  T1 parity of mean/std vs the oracle and of the costs of the
    device's own samples vs the oracle's cost function."""
    L = amd._lib
    N = 2048
    img, truth = orc.synth_sinusoid_image(N, 0)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    assert np.array_equal(grad, orc.comp_grad_img(img, orc.kernel_builder((11, 5))))
    init = truth[[0, -1], :][:, [1, 0]]
    rng = np.random.default_rng(0)
    cols = np.sort(rng.choice(np.arange(1, N - 1), size=1498, replace=False))
    obs = np.stack([cols, truth[cols, 0] + rng.integers(-2, 3, size=cols.size)], axis=1).astype(np.int64)
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 300, 'length_scale': 80}, noise_y=1, N_samples=4000,
              score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, seed=1, fix_endpoints=True)
    tr = amd.GP_Edge_Tracing(init, grad, obs=obs, **kw, _ctx=ctx)
    b = tr._batch
    b.set_obs(0, obs)
    b.fit_predict(want_cov=True)
    p = orc.resolve_params(init, grad, obs=obs, **kw)
    x, y, w = orc.assemble_training(p["init"], obs, True)
    y_s = np.std(y) + 1.0
    fit = orc.gp_fit(x, y / y_s, w, p["sigma_f"] ** 2 / y_s ** 2, p["length_scale"], "RBF", 2.5, 1, N)
    pred = orc.gp_predict(fit, p["x_grid"].astype(float), want_cov=False)
    assert b.scalars().n == 1500
    np.testing.assert_allclose(b.read(L.BUF_ALPHA), fit["alpha"], rtol=1e-5, atol=1e-7 * np.abs(fit["alpha"]).max())
    np.testing.assert_allclose(b.read(L.BUF_MEAN), pred["mean"], rtol=1e-7)
    np.testing.assert_allclose(b.read(L.BUF_STD), pred["std"], rtol=1e-5, atol=1e-7)
    b.factor()
    A = b.read(L.BUF_FACTOR)
    cov = b.read(L.BUF_COV)
    np.testing.assert_allclose(A.T @ A, cov, rtol=0, atol=1e-9 * np.abs(cov).max())
    b.normals([7])
    b.sample()
    b.score()
    Y = b.read(L.BUF_SAMPLES)
    assert Y.shape == (4000, N)
    # sample statistics against the exact posterior
    np.testing.assert_allclose(Y.mean(axis=0), pred["mean"] * y_s, atol=6 * (pred["std"] * y_s).max() / np.sqrt(4000) + 1e-6)
    costs = b.read(L.BUF_COSTS)
    oc = orc.costs_batch(orc.normalise(grad, (0, 1), np.float64), p["x_grid"], Y.T)
    np.testing.assert_allclose(costs, oc, rtol=1e-9)
    assert np.array_equal(b.read(L.BUF_BEST_IDX), np.argsort(oc, kind="stable")[:400])


def test_blocked_fit_variants_agree(amd, ctx):
    """More than 128 training points: alpha by one workgroup per 64-row block (k_chol_solve_mw, published blocks) against the
    single-workgroup solve, and the diagonal blocks factored inside the trailing update against a launch of their own --
    the same factor bit for bit, alpha / mean / std to rounding (sklearn_gpr.py:304-320, 381-436)."""
    L = amd._lib
    N = 1024
    img, truth = orc.synth_sinusoid_image(N, 2)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    rng = np.random.default_rng(1)
    cols = np.sort(rng.choice(np.arange(1, N - 1), size=437, replace=False))  # 439 training points: 7 blocks, a short last one
    obs = np.stack([cols, truth[cols, 0] + rng.integers(-2, 3, size=cols.size)], axis=1).astype(np.int64)
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 150, 'length_scale': 40}, noise_y=1, N_samples=256,
              score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, seed=1, fix_endpoints=True)
    tr = amd.GP_Edge_Tracing(init, grad, obs=obs, **kw, _ctx=ctx)
    b = tr._batch
    res = {}
    for mw, dis in ((1, 1), (0, 1), (1, 0), (0, 0)):
        old = (L.set_option("solve_mw", mw), L.set_option("diag_in_syrk", dis))
        try:
            b.set_obs(0, obs)
            b.fit_predict(want_cov=False)
            n = b.scalars().n
            res[(mw, dis)] = (b.read(L.BUF_CHOL), b.read(L.BUF_ALPHA), b.read(L.BUF_MEAN), b.read(L.BUF_STD))
        finally:
            L.set_option("solve_mw", old[0])
            L.set_option("diag_in_syrk", old[1])
    assert n == 439
    ref = res[(0, 0)]
    for key, (chol, alpha, mean, std) in res.items():
        assert np.array_equal(np.tril(chol), np.tril(ref[0])), key
        np.testing.assert_allclose(alpha, ref[1], rtol=1e-9, atol=1e-11 * np.abs(ref[1]).max(), err_msg=str(key))
        np.testing.assert_allclose(mean, ref[2], rtol=1e-10, err_msg=str(key))
        np.testing.assert_allclose(std, ref[3], rtol=1e-9, atol=1e-12, err_msg=str(key))
    assert np.array_equal(res[(1, 1)][1], res[(1, 0)][1]) and np.array_equal(res[(0, 1)][1], res[(0, 0)][1])


def test_config5_matern_frame_with_warm_start(amd, ctx):
    """1024^2 frame, Matern-5/2 (sigma_f ~ 154, l ~ 41), warm start from every 16th pixel of a
    previous trace (fewer than algo_thresh points): the whole device trace vs the oracle run with
    the library's sign convention -- observation sets and edge trace bit-exact."""
    N = 1024
    img, truth = orc.synth_sinusoid_image(N, 5)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    warm = truth[16:-16:16][:, [1, 0]].astype(np.int64)
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 154, 'length_scale': 41}, noise_y=1,
              N_samples=300, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, seed=3, fix_endpoints=True)
    rec = []
    et_o, ci_o, info = orc.trace(init, grad, obs=warm, record=rec, sign_convention="harmonic", **kw)
    tr = amd.GP_Edge_Tracing(init, grad, obs=warm, **kw, _ctx=ctx)
    et, (all_samples, all_obs, curves) = tr(return_lines=True)
    assert tr._n_iter == info["n_iter"] >= 1
    for i, r in enumerate(rec):
        assert np.array_equal(all_obs[i + 1], r["obs_out"]), "iteration %d" % i
    assert np.array_equal(et, et_o)


def test_large_n_structured_path_equals_generic(amd, ctx):
    """900 on-grid observations on a 2048-column edge (delta_x = 2 keeps the trace unfinished): the structured loop
    path for many training points (blocked fit in HBM, U = L^-1 B by blocked substitution through HBM, H from U)
    must give the factor and mean of the generic path (predict -> covariance -> pivoted Cholesky -> Gram -> Jacobi)."""
    L = amd._lib
    N = 2048
    img, truth = orc.synth_sinusoid_image(N, 0)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    rng = np.random.default_rng(1)
    cols = np.sort(rng.choice(np.arange(1, N - 1), size=900, replace=False))
    obs = np.stack([cols, truth[cols, 0] + rng.integers(-2, 3, size=cols.size)], axis=1).astype(np.int64)
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 300, 'length_scale': 80}, noise_y=1, N_samples=200,
              score_thresh=1, delta_x=2, keep_ratio=0.1, pixel_thresh=5, seed=1, fix_endpoints=True)
    tr = amd.GP_Edge_Tracing(init, grad, **kw, _ctx=ctx)
    b = tr._batch
    info = b.info()
    assert info["structured"] == 1 and info["n_cap"] > 1000
    b.set_obs(0, obs)
    b.fit_predict(want_cov=True)
    b.factor()
    A_gen, ev_gen, mean_gen, cov = b.read(L.BUF_FACTOR), b.read(L.BUF_EIGVALS), b.read(L.BUF_MEAN), b.read(L.BUF_COV)
    alpha_gen = b.read(L.BUF_ALPHA)
    b.set_obs(0, obs)
    for stage in (120, 121, 122, 123):
        b.profile_stage(stage, 1)
    assert b.scalars().n == 902
    np.testing.assert_allclose(b.read(L.BUF_ALPHA), alpha_gen, rtol=1e-12)
    np.testing.assert_allclose(b.read(L.BUF_MEAN), mean_gen, rtol=1e-9)
    A_str = b.read(L.BUF_FACTOR)
    scale = ev_gen[0]
    np.testing.assert_allclose(A_str.T @ A_str, cov, rtol=0, atol=1e-9 * scale)
    # against the oracle too (T1 on the mean)
    p = orc.resolve_params(init, grad, obs=obs, **kw)
    x, y, w = orc.assemble_training(p["init"], obs, True)
    y_s = np.std(y) + 1.0
    fit = orc.gp_fit(x, y / y_s, w, p["sigma_f"] ** 2 / y_s ** 2, p["length_scale"], "RBF", 2.5, 1, N)
    pred = orc.gp_predict(fit, p["x_grid"].astype(float), want_cov=False)
    np.testing.assert_allclose(b.read(L.BUF_MEAN), pred["mean"], rtol=1e-7)


def _matern_frame(amd, ctx, N=1024, S=300):
    img, truth = orc.synth_sinusoid_image(N, 5)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    warm = truth[16:-16:16][:, [1, 0]].astype(np.int64)
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1,
              N_samples=S, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, seed=3, fix_endpoints=True)
    return init, grad, warm, kw


def test_any_rank_factor_full_rank_matern_vs_lapack(amd, ctx):
    """The default factor of a full-rank posterior (pivoted Cholesky over the GPU + one-sided block Jacobi on its
    rows, gpet_eig.hip) on a 1024-wide Matern-5/2 covariance: reconstruction, LAPACK's eigenvalues, orthogonal rows,
    and -- what the tracer sees -- the samples Z A against Z F with F = LAPACK's sqrt(s) v rows (same sign convention), in
    pixels.  (LAPACK on the host is this solver's independent cross-check.)"""
    L = amd._lib
    init, grad, warm, kw = _matern_frame(amd, ctx)
    tr = amd.GP_Edge_Tracing(init, grad, obs=warm, **kw, _ctx=ctx)
    b = tr._batch
    b.set_obs(0, warm)
    b.fit_predict(want_cov=True)
    b.factor()
    s = b.scalars()
    A, cov, ev = b.read(L.BUF_FACTOR), b.read(L.BUF_COV), b.read(L.BUF_EIGVALS)
    assert s.rank == 1024 and 4 <= int(s.lml) <= 14, (s.rank, s.lml)  # (lml field: Jacobi sweeps)
    w = np.linalg.eigvalsh(cov)[::-1]
    assert np.abs(A.T @ A - cov).max() < 2e-12 * np.abs(cov).max()
    assert np.abs(ev - w).max() < 1e-12 * w[0]
    assert (np.abs(ev - w) / w).max() < 1e-6  # Jacobi on the Cholesky factor: small eigenvalues to high RELATIVE accuracy
    Gm = A @ A.T
    d = np.sqrt(np.diag(Gm))
    assert np.abs(Gm / d[:, None] / d[None, :] - np.eye(len(d))).max() < 1e-10
    F, _, _ = orc.mvn_factor_svd(cov, "harmonic")
    Z = orc.legacy_standard_normal(11, 64 * 1024).reshape(64, 1024)
    y_s = s.y_s
    d_lapack = np.abs(Z @ A - Z @ F).max() * y_s
    print("max sample difference against LAPACK's factor: %.3g px" % d_lapack)
    assert d_lapack < 1e-5  # (measured: 2.7e-7 px; round 1's whole-GPU scalar Jacobi, removed in round 5, was 4e-3 px from LAPACK)


@pytest.mark.parametrize("opts", [{}, {"pcx_one_pivot": 1}, {"oj_stage": 0}, {"oj_args": 0}, {"oj_persist": 0},
                                  {"pcx_one_pivot": 1, "oj_stage": 0, "oj_args": 0}])
def test_any_rank_factor_code_paths_agree(amd, ctx, opts):
    """Every variant of the any-rank factor -- blocked or one-pivot-per-launch Cholesky, rows staged in LDS or fed from
    registers, per-edge pointers in the kernel arguments or in the edge table (the forms wide edges and big batches
    take) -- on a batch of three 320-column Matern edges of different observation sets: valid factors (reconstruction,
    LAPACK's eigenvalues) whose samples agree with LAPACK's to 1e-5 px."""
    L = amd._lib
    old = {k: L.set_option(k, v) for k, v in opts.items()}
    try:
        _any_rank_paths_body(amd, ctx, L)
    finally:
        for k, v in old.items():
            L.set_option(k, v)


def _any_rank_paths_body(amd, ctx, L):
    N = 320
    img, truth = orc.synth_sinusoid_image(N, 5)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 48, 'length_scale': 13}, noise_y=1, N_samples=128,
              score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    obs = [truth[s:-s:s][:, [1, 0]].astype(np.int64) for s in (16, 40, 64)]
    bt = amd.GP_Edge_Tracing_Batch([init] * 3, grad, [3, 4, 5], obs=obs, **kw, _ctx=ctx)
    b = bt._batch
    b.fit_predict(want_cov=True)
    b.factor()
    Z = orc.legacy_standard_normal(11, 16 * N).reshape(16, N)
    for e in range(3):
        s = b.scalars(e)
        A, cov, ev = b.read(L.BUF_FACTOR, e), b.read(L.BUF_COV, e), b.read(L.BUF_EIGVALS, e)
        assert s.rank == N
        w = np.linalg.eigvalsh(cov)[::-1]
        assert np.abs(A.T @ A - cov).max() < 2e-12 * np.abs(cov).max()
        assert np.abs(ev - w).max() < 1e-12 * w[0]
        F, _, _ = orc.mvn_factor_svd(cov, "harmonic")
        d = np.abs(Z @ A - Z @ F) * s.y_s
        # the two pinned end points (noise 1e-7 + 1e-6) are a near-degenerate pair of tiny eigenvalues whose
        # eigenvectors are arbitrary within their plane in any solver, LAPACK included: ~6 sigma of that amplitude at
        # the ends, decaying inwards over a few length scales; away from them the samples agree to 1e-5 px
        assert d[:, N // 4:3 * N // 4].max() < 1e-5
        assert d.max() < 6 * np.sqrt(1.2e-6) * s.y_s * np.abs(Z).max()


def test_any_rank_factor_rank_deficient_rbf(amd, ctx):
    """RBF with a short length scale on 1024 columns: numerical rank ~270 -- above the LDS solver's 96, far below Lg.
    The pivoted Cholesky stops at its tolerance and the Jacobi runs on the rows it produced."""
    L = amd._lib
    N = 1024
    img, truth = orc.synth_sinusoid_image(N, 5)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    warm = truth[16:-16:16][:, [1, 0]].astype(np.int64)
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 150, 'length_scale': 10}, noise_y=1, N_samples=300,
              score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, seed=3, fix_endpoints=True)
    tr = amd.GP_Edge_Tracing(init, grad, obs=warm, **kw, _ctx=ctx)
    b = tr._batch
    assert b.info()["factor_cap"] > 96
    b.set_obs(0, warm)
    b.fit_predict(want_cov=True)
    b.factor()
    s = b.scalars()
    A, cov, ev = b.read(L.BUF_FACTOR), b.read(L.BUF_COV), b.read(L.BUF_EIGVALS)
    assert 96 < s.rank < 400, s.rank
    w = np.linalg.eigvalsh(cov)[::-1]
    assert np.abs(A.T @ A - cov).max() < 1e-11 * np.abs(cov).max()
    assert np.abs(ev - w[:s.rank]).max() < 1e-11 * w[0]
    F, _, _ = orc.mvn_factor_svd(cov, "harmonic")
    Z = orc.legacy_standard_normal(11, 64 * 1024).reshape(64, 1024)
    d = np.abs(Z[:, :s.rank] @ A - Z @ F).max() * s.y_s
    print("rank %d, %d sweeps, max sample difference vs LAPACK %.3g px" % (s.rank, int(s.lml), d))
    assert d < 1e-4


def test_config4_batch_equals_single_edge_runs_at_bench_shape(amd, ctx):
    """BASELINE config 4's shape (a batch of independent 500x500 edges, here 48 with different seeds): the edges
    finish after different numbers of iterations, so the loop is advanced in groups over a compacted edge table, and
    the batch's 624 converged-fit problems per launch take the two-tiles-per-thread objective kernel while a single
    edge's 13 take the one-tile one.  Every edge must get what its own single-edge run gets: the same number of
    iterations, the same observations, the same trace; the credible interval to 1e-6 (the two objective kernels add
    their final sums in a different order)."""
    img, truth = orc.synth_sinusoid_image(500, 3)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, noise_y=1, N_samples=1000,
              score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True, return_std=True)
    B = 48
    seeds = [101 + e for e in range(B)]
    batch = amd.GP_Edge_Tracing_Batch([init] * B, grad, seeds, **kw, _ctx=ctx)
    out = batch()
    iters_b = batch._iters()
    obs_b = batch._batch.read_obs_all()
    assert len(set(iters_b)) > 1  # staggered finishing: the compaction is exercised
    for e in range(0, B, 5):
        single = amd.GP_Edge_Tracing(init, grad, seed=seeds[e], **kw, _ctx=ctx)
        et, ci = single()
        assert single._n_iter == iters_b[e], "edge %d" % e
        assert np.array_equal(single._batch.read(amd._lib.BUF_OBS), obs_b[e]), "edge %d" % e
        assert np.array_equal(out[e][0], et), "edge %d" % e
        np.testing.assert_allclose(out[e][1][0], ci[0], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out[e][1][1], ci[1], rtol=1e-6, atol=1e-6)


def test_lds_jacobi_forms_agree_over_ranks(amd, ctx):
    """The two LDS Jacobi kernels for ranks <= 96 -- seated with three barriers per round (k_jacobi_seat: variant 0, the
    cross-check) and seated with the rotation parameters one round ahead and one barrier per round (k_jacobi_ahead: 1, the
    default) -- run the same rotations in the same order, so their factors agree to rounding
    (not bitwise: a pair's roles can be swapped, which changes the order of the additions).  Generic path (stage API:
    covariance -> pivoted Cholesky -> Gram -> Jacobi -> rows) over widths and length scales that give odd, even, tiny and
    capacity-sized ranks; then whole traces through the structured loop path, which must not move by a pixel."""
    L = amd._lib
    seen = set()
    for N, ell, n_obs in [(40, 80.0, 0), (40, 30.0, 0), (64, 6.0, 4), (96, 7.0, 7), (128, 8.0, 10), (200, 9.0, 9), (256, 9.0, 12), (500, 20.0, 30)]:
        img, truth = orc.synth_sinusoid_image(N, 2)
        grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
        init = truth[[0, -1], :][:, [1, 0]]
        if ell == 80.0:  # a 7-column edge: ranks of a handful
            init = truth[[16, 22], :][:, [1, 0]]
        kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 0.15 * N, 'length_scale': ell}, noise_y=1, N_samples=64,
                  score_thresh=1, delta_x=3, keep_ratio=0.1, pixel_thresh=3, seed=1, fix_endpoints=True)
        out = {}
        for variant in (0, 1):
            old = L.set_option("jacobi_variant", variant)
            try:
                tr = amd.GP_Edge_Tracing(init, grad, **kw, _ctx=ctx)
                b = tr._batch
                if n_obs:
                    cols = np.linspace(2, N - 3, n_obs).astype(np.int64)
                    b.set_obs(0, np.stack([cols, truth[cols, 0]], axis=1).astype(np.int64))
                b.fit_predict(want_cov=True)
                b.factor()
                out[variant] = (b.read(L.BUF_FACTOR), b.read(L.BUF_EIGVALS), b.read(L.BUF_COV), b.scalars().rank)
            finally:
                L.set_option("jacobi_variant", old)
        A0, ev0, cov, r0 = out[0]
        A1, ev1, _, r1 = out[1]
        assert r0 == r1 and A0.shape == A1.shape and r0 <= 96  # (above 96 the any-rank factor of csrc/gpet_eig.hip takes over)
        seen.add(r0)
        scale = ev0[0]
        np.testing.assert_allclose(ev1, ev0, rtol=0, atol=1e-12 * scale)
        for A_ in (A0, A1):
            np.testing.assert_allclose(A_.T @ A_, cov, rtol=0, atol=1e-9 * scale)
        # the same rotations in the same order, one round-off apart: rows of well separated singular values agree row by row
        # (a near-degenerate pair may come out rotated within its plane)
        gap = np.minimum(np.abs(np.diff(ev0, prepend=np.inf)), np.abs(np.diff(ev0, append=-np.inf)))
        sep = gap > 1e-6 * scale
        np.testing.assert_allclose(A1[sep], A0[sep], rtol=0, atol=1e-10 * np.sqrt(scale))
    assert any(r % 2 for r in seen) and any(r % 2 == 0 for r in seen) and min(seen) <= 8 and max(seen) >= 60, seen
    # whole traces, structured loop path (the bench configuration at 128 columns and at the README size)
    for N, S, ell, sf in [(128, 200, 8.0, 20.0), (500, 1000, 20.0, 75.0)]:
        img, truth = orc.synth_sinusoid_image(N, 3)
        grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
        init = truth[[0, -1], :][:, [1, 0]]
        kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': sf, 'length_scale': ell}, noise_y=1, N_samples=S,
                  score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
        traces = {}
        for variant in (0, 1):
            old = L.set_option("jacobi_variant", variant)
            try:
                bt = amd.GP_Edge_Tracing_Batch([init] * 4, grad, [1, 2, 3, 4], **kw, _ctx=ctx)
                assert bt._batch.info()["structured"] == 1
                traces[variant] = bt()
            finally:
                L.set_option("jacobi_variant", old)
        for a, b_ in zip(traces[0], traces[1]):
            assert np.array_equal(a, b_)


def test_structured_loop_warm_started_eigen_decomposition(amd, ctx):
    """Option jacobi_warm (k_jacobi_prerot): from the second iteration on the eigen-decomposition of the structured path
    starts from the previous iteration's eigenvectors.  Same traces as the cold start (the factor is the same matrix's, to
    rounding), fewer Jacobi sweeps, and repeated launches on a fixed state read a slot they do not write."""
    L = amd._lib
    img, truth = orc.synth_sinusoid_image(160, 2)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 30, 'length_scale': 8}, noise_y=1, N_samples=256, score_thresh=1,
              delta_x=5, keep_ratio=0.1, pixel_thresh=4, fix_endpoints=True)
    E = 40  # (above jlog_max_b: the LDS form of W; 4 edges below: the rotation log + k_jacobi_wpass)
    for nb in (E, 4):
        seeds = [5 + 997 * k for k in range(nb)]
        res = {}
        for warm in (0, 1):
            old = L.set_option("jacobi_warm", warm)
            try:
                bt = amd.GP_Edge_Tracing_Batch([init] * nb, grad, seeds, **kw, _ctx=ctx)
                assert bt._batch.info()["structured"] == 1
                bt._batch.iterate(seeds, 3)
                sweeps = [int(bt._batch.scalars(e).lml) for e in range(nb)]
                f1 = np.array(bt._batch.read(L.BUF_FACTOR, 0))
                # the factor stage again on the same state: now from the eigenvectors iteration 2 left (slot 0 of the ring;
                # these launches write slot 1): the same matrix's factor, and every repetition the same bits
                bt._batch.profile_stage(1, 1)
                f2 = np.array(bt._batch.read(L.BUF_FACTOR, 0))
                bt._batch.profile_stage(1, 2)
                f3 = np.array(bt._batch.read(L.BUF_FACTOR, 0))
                bt._batch.close()
                traces = amd.GP_Edge_Tracing_Batch([init] * nb, grad, seeds, **kw, _ctx=ctx)()
            finally:
                L.set_option("jacobi_warm", old)
            assert np.array_equal(f2, f3)
            assert np.max(np.abs(f1 - f2)) <= 1e-7 * np.max(np.abs(f1))
            res[warm] = (sweeps, f1, traces)
        assert all(np.array_equal(a, b) for a, b in zip(res[0][2], res[1][2]))
        assert np.max(np.abs(res[0][1] - res[1][1])) <= 1e-7 * np.max(np.abs(res[0][1]))
        assert np.mean(res[1][0]) < np.mean(res[0][0]), (res[0][0], res[1][0])


def test_any_rank_factor_warm_start(amd, ctx):
    """Option oj_warm (k_ojw_*): from the second iteration on the one-sided Jacobi of a full-rank (Matern) covariance starts
    from rows built out of the previous iteration's factor instead of a pivoted Cholesky factor.  A^T A = Sigma to rounding
    either way, rows within the solver's tolerance of the cold start's, the same observation sets, fewer sweeps.  Across
    traces only where the caller says the images continue a sequence (set_frame(next_frame=True) =
    gpet_batch_set_images with GPET_IMAGES_NEXT_FRAME): the first factor of the new trace then starts warm; after a
    plain reset() it starts cold -- exactly the sweeps of a fresh object."""
    L = amd._lib
    N = 256
    img, truth = orc.synth_sinusoid_image(N, 5)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1,
              N_samples=200, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    res = {}
    for warm in (0, 1):
        old = L.set_option("oj_warm", warm)
        try:
            bt = amd.GP_Edge_Tracing_Batch([init] * 2, grad, [3, 4], **kw, _ctx=ctx)
            b = bt._batch
            b.iterate([3, 4], 1)
            first_fresh = int(b.scalars(0).lml)  # sweeps of a trace's first factor on a fresh object: always cold
            b.iterate([3, 4], 2)
            b.profile_stage(0, 1)  # covariance of the current observation set
            b.profile_stage(1, 1)  # its factor (previous rows: iteration 2's)
            s = b.scalars(0)
            A = np.array(b.read(L.BUF_FACTOR, 0))
            cov = np.array(b.read(L.BUF_COV, 0))
            obs = b.read_obs_all()
            first_reset = first_frame = None
            if warm:
                b.reset()  # an unrelated second trace: nothing carried over
                b.iterate([3, 4], 1)
                first_reset = int(b.scalars(0).lml)
                b.iterate([3, 4], 2)
                bt.set_frame(grad, None, [3, 4], next_frame=True)  # the next frame of a sequence: first factor warm
                b.iterate([3, 4], 1)
                first_frame = int(b.scalars(0).lml)
            b.close()
        finally:
            L.set_option("oj_warm", old)
        assert s.status == 0 and s.rank == N
        assert np.max(np.abs(A.T @ A - cov)) <= 1e-9 * np.max(np.abs(cov))
        res[warm] = (int(s.lml), A, obs, first_fresh, first_reset, first_frame)
    assert res[1][0] < res[0][0], (res[0][0], res[1][0])
    assert np.max(np.abs(res[1][1] - res[0][1])) <= 1e-6 * np.max(np.abs(res[0][1]))
    assert all(np.array_equal(a, c) for a, c in zip(res[1][2], res[0][2]))
    assert res[1][3] == res[0][3] == res[1][4], res  # fresh object and reset object: the cold start's sweeps
    assert res[1][5] < res[1][3], res                 # next frame: warm


def test_matern_trace_on_a_reused_batch_equals_a_fresh_object(amd, ctx):
    """A trace is a function of (image, seed, observations) only: a Matern batch object that has just traced image A and
    is then given image B (set_frame(next_frame=False)) or reset() must produce, bit for bit, the observation sets,
    iteration counts and traces of a FRESH object on B -- and the oracle's.  (Round 4's default carried the last trace's
    factor rows into the next trace's first factorisation.)"""
    N = 128
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1,
              N_samples=200, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    grads, inits = [], []
    for seed in (5, 9):
        img, truth = orc.synth_sinusoid_image(N, seed)
        grads.append(amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx))
        inits.append(truth[[0, -1], :][:, [1, 0]])
    init = inits[0]
    seeds = [3, 4]

    def run(bt):
        traces = bt()
        return [np.array(t) for t in traces], list(bt.timings["iters"]), [np.array(o) for o in bt._batch.read_obs_all()]

    fresh = amd.GP_Edge_Tracing_Batch([init] * 2, grads[1], seeds, **kw, _ctx=ctx)
    want = run(fresh)
    fresh._batch.close()
    used = amd.GP_Edge_Tracing_Batch([init] * 2, grads[0], [11, 12], **kw, _ctx=ctx)
    run(used)  # image A, other seeds
    used.set_frame(grads[1], None, seeds, next_frame=False)
    got = run(used)
    used.reset()
    again = run(used)
    used._batch.close()
    for res in (got, again):
        assert res[1] == want[1]
        assert all(np.array_equal(a, b) for a, b in zip(res[0], want[0]))
        assert all(np.array_equal(a, b) for a, b in zip(res[2], want[2]))
    for e in range(2):
        et_o, _, info = orc.trace(init, grads[1], seed=seeds[e], sign_convention="harmonic", **kw)
        assert info["n_iter"] == want[1][e]
        assert np.array_equal(want[0][e], et_o)


def test_any_rank_factor_warm_start_falls_back_to_the_pivoted_cholesky(amd, ctx):
    """A non-positive pivot in the warm start's Cholesky (injected: option oj_warm_fail) must leave the factor to the
    pivoted Cholesky, whose launches are enqueued anyway: exactly the cold start's rows."""
    L = amd._lib
    N = 192
    img, truth = orc.synth_sinusoid_image(N, 5)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1,
              N_samples=200, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    rows = {}
    for name, opts in (("cold", {"oj_warm": 0}), ("failed warm start", {"oj_warm": 1, "oj_warm_fail": 1})):
        old = {k: L.set_option(k, v) for k, v in opts.items()}
        try:
            bt = amd.GP_Edge_Tracing_Batch([init], grad, [3], **kw, _ctx=ctx)
            bt._batch.iterate([3], 3)
            s = bt._batch.scalars(0)
            rows[name] = (np.array(bt._batch.read(L.BUF_FACTOR, 0)), int(s.lml), s.status, bt._batch.read_obs_all())
            bt._batch.close()
        finally:
            for k, v in old.items():
                L.set_option(k, v)
    a, b = rows["cold"], rows["failed warm start"]
    assert b[2] == 0 and a[1] == b[1]
    assert np.array_equal(a[0], b[0])
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))


@pytest.mark.parametrize("pchol_multi", [0, 1, 2])
def test_pivoted_cholesky_over_the_gpu_in_a_batch_of_mixed_widths(amd, ctx, pchol_multi):
    """The multi-workgroup pivoted Cholesky (k_pcb_block / k_pcx_step) is picked per BATCH from its widest edge and then runs on
    every edge, writing each edge's transposed copy Gt: a narrower edge beside a > 1 024-column one must own a full Gt (round 5
    sized it from the edge's own width: out-of-bounds writes over the narrow edge's neighbours).  Generic factor path
    (struct_path = 0, stage API): every edge's factor reproduces its covariance and equals its single-edge run."""
    L = amd._lib
    N, M = 2048, 192
    grad = np.random.default_rng(0).random((M, N)).astype(np.float32)  # (only the GP stages run: any image does)
    yc = lambda x: int(round(96 + 60 * np.sin(x / 200.0)))
    spans = [(0, N - 1), (100, 700)]
    inits = [np.array([[a, yc(a)], [b, yc(b)]]) for a, b in spans]
    rng = np.random.default_rng(5)
    obs = []
    for (a, b), k in zip(spans, (420, 70)):
        cols = np.sort(rng.choice(np.arange(a + 1, b), size=k, replace=False))
        obs.append(np.stack([cols, [yc(c) for c in cols]], axis=1).astype(np.int64))
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 40, 'length_scale': 120}, noise_y=1, N_samples=128,
              score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    old = {"struct_path": L.set_option("struct_path", 0), "pchol_multi": L.set_option("pchol_multi", pchol_multi)}
    try:
        bt = amd.GP_Edge_Tracing_Batch(inits, grad, [1, 2], **kw, obs=obs, _ctx=ctx)
        b = bt._batch
        assert b.info()["structured"] == 0
        for e in range(2):
            b.set_obs(e, obs[e])
        b.fit_predict(want_cov=True)
        b.factor()
        for e in range(2):
            one = amd.GP_Edge_Tracing(inits[e], grad, seed=1 + e, **kw, obs=obs[e], _ctx=ctx)._batch
            one.set_obs(0, obs[e])
            one.fit_predict(want_cov=True)
            one.factor()
            A, cov = b.read(L.BUF_FACTOR, e), b.read(L.BUF_COV, e)
            # (the single-edge run of the narrow edge takes the one-workgroup Cholesky: another pivot order and another truncated
            #  tail -- the numerical rank may differ by a direction or two of ~1e-14 of the largest, the rows of the directions
            #  that matter are the same)
            assert b.scalars(e).status == 0 and abs(b.scalars(e).rank - one.scalars().rank) <= 3
            np.testing.assert_allclose(A.T @ A, cov, rtol=0, atol=1e-9 * np.abs(cov).max(), err_msg="edge %d" % e)
            np.testing.assert_array_equal(cov, one.read(L.BUF_COV))
            A1, ev = one.read(L.BUF_FACTOR), b.read(L.BUF_EIGVALS, e)
            k = min(A.shape[0], A1.shape[0])
            k = int(np.sum(ev[:k] > 1e-9 * ev[0]))
            np.testing.assert_allclose(A[:k], A1[:k], rtol=0, atol=1e-6 * np.abs(A).max(), err_msg="edge %d vs single" % e)
            one.close()
        b.close()
    finally:
        for k, v in old.items():
            L.set_option(k, v)


def test_any_rank_jacobi_half_panel_staging_is_bit_identical(amd, ctx):
    """k_oj_persist stages a pair's 16-row panel whole (131 KB of LDS at 1 024 columns: one workgroup per CU) or one 512-column
    half at a time (66 KB: two per CU; the automatic choice once a round has more pair slots than the GPU has CUs -- config 5's
    eight chains).  The Gram matrix accumulates its chunks in the same order either way and the row update is chunk-wise: the
    factor rows must be the same BITS, on a single edge (whole by default) and on a batch of five (half by default)."""
    L = amd._lib
    init, grad, warm, kw = _matern_frame(amd, ctx)
    kw = {k: v for k, v in kw.items() if k != "seed"}  # (the batch form takes one seed per edge)
    for nb in (1, 5):
        rows = {}
        for half in (0, 1):
            old = L.set_option("oj_half_stage", half)
            try:
                bt = amd.GP_Edge_Tracing_Batch([init] * nb, grad, [3 + e for e in range(nb)], obs=[warm] * nb, **kw, _ctx=ctx)
                b = bt._batch
                b.fit_predict(want_cov=True)
                b.factor()
                rows[half] = [np.array(b.read(L.BUF_FACTOR, e)) for e in range(nb)]
                assert all(b.scalars(e).status == 0 and b.scalars(e).rank == 1024 for e in range(nb))
                b.close()
            finally:
                L.set_option("oj_half_stage", -1 if old == 2 else old)
        for e in range(nb):
            assert np.array_equal(rows[0][e], rows[1][e]), (nb, e)
