"""GPU parity of the per-stage hot path (a1-a7) through the C ABI, against the golden
fixtures produced by the unmodified reference and against the CPU oracle."""
import numpy as np
import pytest

from oracle import gpet_oracle as orc
from tests.test_oracle_vs_golden import CTOR, STAGES

pytestmark = pytest.mark.gpu

REL = 1e-5  # north_star tolerance on posterior mean/std


@pytest.fixture(scope="module")
def amd():
    import gaussian_process_edge_trace_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def ctx(amd):
    return amd._lib.Context(0)


def _tracer(amd, ctx, g, name, **extra):
    return amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **CTOR[name], _ctx=ctx, **extra)


@pytest.mark.parametrize("name", ["stage_rbf64", "stage_rbf65", "stage_mat128", "stage_mat15_96", "stage_mat35_96"])
def test_grad_image_bit_exact(amd, ctx, golden, name):
    g = golden(name)
    out = amd.gpet_utils.comp_grad_img(g["in_img"], g["in_kernel"], ctx=ctx)
    assert out.dtype == np.float32
    assert np.array_equal(out, g["ref_grad"])


def test_grad_image_500_and_even_kernels(amd, ctx, golden):
    img, _ = orc.synth_sinusoid_image(500, 1)
    out = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    assert np.array_equal(out, golden("stage_rbf500")["ref_grad"])
    rng = np.random.default_rng(0)
    small = rng.normal(size=(37, 53))
    for ks in [(4, 4), (3, 6), (2, 5), (1, 1), (7, 1)]:
        kk = rng.normal(size=ks)
        assert np.array_equal(amd.gpet_utils.comp_grad_img(small, kk, ctx=ctx), orc.comp_grad_img(small, kk))


@pytest.mark.parametrize("name", STAGES)
def test_ctor_normalised_grad(amd, ctx, golden, name):
    g = golden(name)
    tr = _tracer(amd, ctx, g, name)
    assert np.array_equal(tr.grad_img, g["ref_grad64"].astype(np.float64))
    assert [tr.x_st, tr.x_en, tr.N_samples, tr.N_keep, tr.N_subints, tr.algo_thresh, tr.delta_x, tr.pixel_thresh,
            tr.edge_length] == list(g["ref_scalars"])


@pytest.mark.parametrize("name", STAGES)
def test_fit_predict_T1(amd, ctx, golden, name):
    """T1: deterministic quantities K/L/alpha/mean/std/cov within 1e-5 rel of the reference."""
    g = golden(name)
    L = amd._lib
    tr = _tracer(amd, ctx, g, name)
    b = tr._batch
    b.set_obs(0, g["in_obs"])
    b.fit_predict(want_cov=True)
    s = b.scalars()
    assert s.n == g["ref_X_train"].shape[0]
    assert np.array_equal(b.read(L.BUF_X_TRAIN), g["ref_X_train"])
    np.testing.assert_allclose(b.read(L.BUF_Y_TRAIN), g["ref_y_train"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(s.amp, float(g["ref_amp"]), rtol=1e-14)
    np.testing.assert_allclose(s.y_mean, float(g["ref_y_train_mean"]), rtol=1e-13)
    np.testing.assert_allclose(s.y_std, float(g["ref_y_train_std"]), rtol=1e-13)
    np.testing.assert_allclose(b.read(L.BUF_CHOL), g["ref_L"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b.read(L.BUF_ALPHA), g["ref_alpha"], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(b.read(L.BUF_MEAN), g["ref_mean"], rtol=REL * 1e-3)
    np.testing.assert_allclose(b.read(L.BUF_STD), g["ref_std"], rtol=REL, atol=1e-8)
    cov = b.read(L.BUF_COV)
    scale = np.abs(g["ref_cov_diag"]).max()
    np.testing.assert_allclose(np.diag(cov), g["ref_cov_diag"], rtol=REL, atol=1e-9 * scale)
    np.testing.assert_allclose(cov[0], g["ref_cov_row0"], rtol=REL, atol=1e-9 * scale)
    np.testing.assert_allclose(cov[cov.shape[0] // 2], g["ref_cov_rowmid"], rtol=REL, atol=1e-9 * scale)
    assert np.array_equal(cov, cov.T)
    if "ref_cov" in g:
        np.testing.assert_allclose(cov, g["ref_cov"], rtol=REL, atol=1e-9 * scale)


@pytest.mark.parametrize("name", STAGES)
def test_normals_stream(amd, ctx, golden, name):
    """Device MT19937 + legacy polar gauss reproduces RandomState(seed).standard_normal."""
    g = golden(name)
    L = amd._lib
    tr = _tracer(amd, ctx, g, name, z_cols=int(g["ref_scalars"][8]))
    b = tr._batch
    b.normals([int(g["in_gp_seed"])])
    Z = b.read(L.BUF_NORMALS)
    S, N = Z.shape
    ref = orc.legacy_standard_normal(int(g["in_gp_seed"]), S * N).reshape(S, N)
    assert np.array_equal(Z[:8], g["ref_Z_head"]) or np.abs(Z[:8] - g["ref_Z_head"]).max() < 1e-14
    # accept/reject decisions must be identical everywhere (else the stream shifts)
    np.testing.assert_allclose(Z, ref, rtol=0, atol=5e-15)


@pytest.mark.parametrize("name,z_cols", [("stage_rbf500", 96), ("stage_rbf500", 250), ("stage_rbf500", 251),
                                         ("stage_rbf65", 16), ("stage_rbf64", 1)])
def test_normals_stream_sparse_columns(amd, ctx, golden, name, z_cols):
    """Only the first z_cols normals of every sample row are kept (the structured loop path needs the factor's rank
    many): the kept ones must be the reference stream's -- the queued evaluation (z_cols <= N/2) and the in-place one
    (above), odd widths where a pair straddles two rows, a single kept column."""
    g = golden(name)
    L = amd._lib
    tr = _tracer(amd, ctx, g, name, z_cols=z_cols)
    b = tr._batch
    seed = int(g["in_gp_seed"])
    b.normals([seed])
    Z = b.read(L.BUF_NORMALS)
    S = Z.shape[0]
    N = int(g["ref_scalars"][8])
    assert Z.shape[1] == z_cols
    ref = orc.legacy_standard_normal(seed, S * N).reshape(S, N)[:, :z_cols]
    np.testing.assert_allclose(Z, ref, rtol=0, atol=5e-15)


@pytest.mark.parametrize("z_cols", [65, 20])
def test_normals_stream_odd_count(amd, ctx, golden, z_cols):
    """151 samples x 65 columns: an odd number of normals, so the last polar pair contributes only its first value."""
    g = golden("stage_rbf65")
    L = amd._lib
    kw = dict(CTOR["stage_rbf65"], N_samples=151)
    tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **kw, _ctx=ctx, z_cols=z_cols)
    b = tr._batch
    seed = 12345
    b.normals([seed])
    Z = b.read(L.BUF_NORMALS)
    assert Z.shape == (151, z_cols)
    ref = orc.legacy_standard_normal(seed, 151 * 65).reshape(151, 65)[:, :z_cols]
    np.testing.assert_allclose(Z, ref, rtol=0, atol=5e-15)


@pytest.mark.parametrize("name,z_cols,n_samples", [("stage_rbf500", 500, 1000), ("stage_rbf500", 96, 1000), ("stage_rbf500", 251, 1000),
                                                   ("stage_rbf65", 65, 4001), ("stage_rbf65", 16, 9000), ("stage_rbf64", 64, 700)])
def test_normals_stream_chunked_equals_sequential(amd, ctx, golden, name, z_cols, n_samples):
    """One long MT19937 stream generated by many workgroups (jump-ahead to the state of every chunk of 64 blocks, accept
    counts, prefix, emit; option rng_chunked=1) must be the sequential generator's stream bit for bit -- and numpy's
    RandomState(seed).standard_normal: full and sparse column sets, odd widths and an odd total (a pair straddling rows /
    ending the stream), streams of 2..33 chunks, the last chunk running on past its 64 blocks."""
    g = golden(name)
    L = amd._lib
    kw = dict(CTOR[name], N_samples=n_samples)
    seed = 987654
    out = {}
    for mode in (0, 1):
        old = L.set_option("rng_chunked", mode)
        try:
            tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **kw, _ctx=ctx, z_cols=z_cols)
            tr._batch.normals([seed])
            out[mode] = tr._batch.read(L.BUF_NORMALS)
        finally:
            L.set_option("rng_chunked", -1 if old == 2 else old)
    assert out[0].shape == (n_samples, z_cols)
    assert np.array_equal(out[0], out[1])
    N = int(g["ref_scalars"][8])
    ref = orc.legacy_standard_normal(seed, n_samples * N).reshape(n_samples, N)[:, :z_cols]
    np.testing.assert_allclose(out[1], ref, rtol=0, atol=5e-15)


@pytest.mark.parametrize("name,z_cols,n_samples", [("stage_rbf500", 72, 1000), ("stage_rbf500", 96, 1000), ("stage_rbf500", 250, 1000),
                                                   ("stage_rbf500", 500, 1000), ("stage_rbf500", 71, 333), ("stage_rbf64", 64, 700),
                                                   ("stage_rbf64", 16, 2000), ("stage_rbf64", 1, 128), ("stage_mat128", 128, 256)])
def test_normals_stream_register_resident_generator(amd, ctx, golden, name, z_cols, n_samples):
    """k_mt_normals4 (csrc/gpet_rng.hip: MT19937 state in registers, four streams per wave, float32 pre-filter of the accept
    test, queued evaluation of the stored pairs; option rng4 = 1 forces it on any launch shape) against the one-workgroup-
    per-stream generator bit for bit and against numpy's RandomState(seed).standard_normal: the sparse form (<= 96 stored
    columns of a wide row), the dense form (everything stored, or a narrow row), odd column counts, a single column."""
    g = golden(name)
    L = amd._lib
    kw = dict(CTOR[name], N_samples=n_samples)
    seed = 424242
    out = {}
    for mode in (0, 1):
        old4, oldc = L.set_option("rng4", mode), L.set_option("rng_chunked", 0)
        try:
            tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **kw, _ctx=ctx, z_cols=z_cols)
            tr._batch.normals([seed])
            out[mode] = tr._batch.read(L.BUF_NORMALS)
        finally:
            L.set_option("rng4", -1 if old4 == 2 else old4)
            L.set_option("rng_chunked", -1 if oldc == 2 else oldc)
    assert out[1].shape == (n_samples, z_cols)
    N = int(g["ref_scalars"][8])
    ref = orc.legacy_standard_normal(seed, n_samples * N).reshape(n_samples, N)[:, :z_cols]
    np.testing.assert_allclose(out[1], ref, rtol=0, atol=5e-15)
    assert np.array_equal(out[0], out[1])


def test_normals_register_resident_generator_batch_ring_and_loop(amd, ctx, golden):
    """The same generator where it runs in production: a batch (streams = edges x iterations ahead, four per wave, the last
    wave partly filled), every ring slot of the loop's look-ahead, finished edges skipped -- whole traces and iteration
    counts equal those of the one-workgroup-per-stream generator."""
    g = golden("stage_rbf500")
    L = amd._lib
    kw = dict(CTOR["stage_rbf500"])
    kw.pop("seed")
    B = 7
    seeds = [5 + 997 * e for e in range(B)]
    res = {}
    for mode in (0, 1):
        old4 = L.set_option("rng4", mode)
        try:
            bt = amd.GP_Edge_Tracing_Batch([g["in_init"]] * B, g["ref_grad"], seeds, **kw, _ctx=ctx)
            assert bt._batch.info()["structured"] == 1
            res[mode] = ([np.asarray(t) for t in bt()], bt.timings["iters"])
            bt._batch.close()
        finally:
            L.set_option("rng4", -1 if old4 == 2 else old4)
    assert res[0][1] == res[1][1]
    for a, c in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, c)


def test_normals_stream_config3_shape(amd, ctx):
    """BASELINE config 3's stream: RandomState(seed).standard_normal((4000, 2048)) = 8.2 M normals of ONE stream (21 M words,
    ~525 chunks, ten doubling levels of the jump-ahead), chosen automatically for a single edge; against numpy's own
    generator (the test may use numpy: the product does not)."""
    L = amd._lib
    N = 2048
    img, truth = orc.synth_sinusoid_image(N, 0)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 300, 'length_scale': 80}, noise_y=1, N_samples=4000,
              score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, seed=1, fix_endpoints=True)
    tr = amd.GP_Edge_Tracing(init, grad, **kw, _ctx=ctx, z_cols=N)
    tr._batch.normals([7])
    Z = tr._batch.read(L.BUF_NORMALS)
    ref = np.random.RandomState(7).standard_normal((4000, N))
    np.testing.assert_allclose(Z, ref, rtol=0, atol=5e-15)


@pytest.mark.parametrize("name", ["stage_rbf64", "stage_rbf65", "stage_mat128", "stage_mat15_96", "stage_mat35_96"])
def test_sample_T2_injected_factor(amd, ctx, golden, name):
    """T2: with the reference's factor sqrt(s)*v injected, samples match the reference."""
    g = golden(name)
    L = amd._lib
    N = int(g["ref_scalars"][8])
    tr = _tracer(amd, ctx, g, name, z_cols=N)
    b = tr._batch
    b.set_obs(0, g["in_obs"])
    b.fit_predict(want_cov=False)
    b.write(L.BUF_FACTOR, g["ref_factor"], rows=N)
    b.normals([int(g["in_gp_seed"])])
    b.sample()
    Y = b.read(L.BUF_SAMPLES).T
    np.testing.assert_allclose(Y, g["ref_samples_head"], rtol=1e-9, atol=1e-8)


@pytest.mark.parametrize("name", STAGES)
def test_eigen_factor_matches_svd_up_to_sign(amd, ctx, golden, name):
    """Production factor: rows sqrt(s_k) v_k of the posterior covariance, k by descending s_k.
    Compared with the reference's LAPACK SVD after aligning the (implementation-defined) signs."""
    g = golden(name)
    L = amd._lib
    tr = _tracer(amd, ctx, g, name)
    b = tr._batch
    b.set_obs(0, g["in_obs"])
    b.fit_predict(want_cov=True)
    b.factor()
    A = b.read(L.BUF_FACTOR)
    ev = b.read(L.BUF_EIGVALS)
    r = A.shape[0]
    s_ref = g["ref_svals"]
    np.testing.assert_allclose(ev[:r], s_ref[:r], rtol=1e-6, atol=1e-10 * s_ref[0])
    # reconstruct the covariance
    cov = b.read(L.BUF_COV)
    np.testing.assert_allclose(A.T @ A, cov, rtol=0, atol=1e-9 * s_ref[0])
    F = g["ref_factor"] if "ref_factor" in g else g["ref_factor_top"]
    k = min(r, F.shape[0])
    sig = s_ref[:k] > 1e-9 * s_ref[0]
    for i in np.nonzero(sig)[0]:
        sgn = np.sign(A[i] @ F[i])
        np.testing.assert_allclose(sgn * A[i], F[i], rtol=0, atol=2e-5 * np.sqrt(s_ref[i]) + 1e-9 * np.sqrt(s_ref[0]))


@pytest.mark.parametrize("name", STAGES)
def test_samples_T2_sign_aligned(amd, ctx, golden, name):
    """Full device path (fit -> factor -> MT normals -> GEMM): equals the reference samples once
    each eigenvector's sign is aligned with the reference SVD's (flip = negate that Z column)."""
    g = golden(name)
    L = amd._lib
    tr = _tracer(amd, ctx, g, name)
    b = tr._batch
    b.set_obs(0, g["in_obs"])
    b.fit_predict(want_cov=True)
    b.factor()
    A = b.read(L.BUF_FACTOR)
    F = g["ref_factor"] if "ref_factor" in g else g["ref_factor_top"]
    k = min(A.shape[0], F.shape[0])
    sgn = np.sign(np.einsum("ij,ij->i", A[:k], F[:k]))
    sgn[sgn == 0] = 1
    b.normals([int(g["in_gp_seed"])])
    Z = b.read(L.BUF_NORMALS)
    Z[:, :k] *= sgn[None, :]
    b.write(L.BUF_NORMALS, Z)
    b.sample()
    Y = b.read(L.BUF_SAMPLES).T
    ns = g["ref_samples_head"].shape[1]
    ref = g["ref_samples_head"]
    # (near-)degenerate singular values have LAPACK-arbitrary singular vectors even in the
    # reference: those directions may differ by up to ~6 sigma of their own amplitude.
    sv = g["ref_svals"]
    gap = np.minimum(np.abs(np.diff(sv, prepend=np.inf)), np.abs(np.diff(sv, append=-np.inf)))
    degenerate = sv[(gap < 1e-4 * sv) & (sv > 0)]
    y_s = b.scalars().y_s
    atol = REL * np.abs(ref).max() + 6.0 * y_s * np.sqrt(degenerate.sum())
    np.testing.assert_allclose(Y[:, :ns], ref, rtol=REL, atol=atol)


@pytest.mark.parametrize("name", STAGES)
def test_scoring_on_reference_samples(amd, ctx, golden, name):
    """a7 with the reference's own samples in: costs to 1e-9, best_idxs bit-exact."""
    g = golden(name)
    L = amd._lib
    tr = _tracer(amd, ctx, g, name)
    b = tr._batch
    ref = g["ref_samples_head"]
    ns = ref.shape[1]
    S = tr.N_samples
    Y = np.zeros((S, tr.edge_length))
    Y[:ns] = ref.T
    if ns < S:  # pad with the oracle's samples (same distribution), scored by the oracle below
        p = orc.resolve_params(g["in_init"], g["ref_grad"], **CTOR[name])
        Yo = orc.fit_predict_samples(p["init"], g["in_obs"], p, int(g["in_gp_seed"]))
        Y[ns:] = Yo.T[ns:]
    b.write(L.BUF_SAMPLES, Y)
    b.score()
    costs = b.read(L.BUF_COSTS)
    np.testing.assert_allclose(costs[:ns], g["ref_costs"][:ns], rtol=1e-9)
    oc = orc.costs_batch(g["ref_grad64"].astype(np.float64), tr.x_grid, Y.T)
    np.testing.assert_allclose(costs, oc, rtol=1e-9)
    idx = b.read(L.BUF_BEST_IDX)
    assert np.array_equal(idx, np.argsort(oc, kind="stable")[:tr.N_keep])
    if ns == S:
        assert np.array_equal(idx, g["ref_best_idxs"])
        np.testing.assert_allclose(b.read(L.BUF_BEST_COSTS), g["ref_best_costs"], rtol=1e-9)


def test_not_pd_reports_error(amd, ctx, golden):
    g = golden("stage_rbf64")
    kw = dict(CTOR["stage_rbf64"])
    kw["fix_endpoints"] = True
    tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **kw, _ctx=ctx)
    b = tr._batch
    # duplicate noiseless points at one x with conflicting y are still PD thanks to jitter; force
    # failure with a negative jitter through the params struct instead
    tr._abi.jitter = -1e9
    b2 = amd._lib.Batch(ctx, [g["ref_grad"]], [tr._abi], [tr.init])
    b2.set_obs(0, g["in_obs"])
    with pytest.raises(amd._lib.GpetError) as ei:
        b2.fit_predict()
    assert ei.value.code == amd._lib.ERR_NOT_PD


def test_gaussian_process_regressor_mirror(amd, ctx, golden):
    """Package export GaussianProcessRegressor (sklearn_gpr.py:183-438 subset): fit + predict mean/std
    with the reference's swapped normalize_y semantics, against the oracle's closed forms."""
    from gaussian_process_edge_trace_amd.sklearn_gpr import GaussianProcessRegressor, WeightedWhiteKernel
    rng = np.random.default_rng(0)
    x = np.sort(rng.choice(np.arange(0, 200), size=30, replace=False)).astype(float)
    y = 5 * np.sin(x / 20.0) + rng.normal(0, 0.3, x.size) + 3.0
    w = np.ones(x.size)
    w[[0, -1]] = 1e-7
    xq = np.arange(0, 200, dtype=float)
    for kt, nu in [("RBF", 2.5), ("Matern", 1.5)]:
        for normalize_y in (True, False):
            kern = dict(kernel=kt, nu=nu, constant=2.5, length_scale=15.0,
                        white=WeightedWhiteKernel(noise_weight=w, edge_length=xq.size, noise_level=0.4))
            gp = GaussianProcessRegressor(kernel=kern, alpha=1e-6, normalize_y=normalize_y, _ctx=ctx).fit(x[:, None], y)
            mean, std = gp.predict(xq[:, None], return_std=True)
            fit = orc.gp_fit(x, y, w, 2.5, 15.0, kt, nu, 0.4, xq.size, jitter=1e-6, center=True,
                             scale=not normalize_y)
            pred = orc.gp_predict(fit, xq, want_cov=False)
            np.testing.assert_allclose(mean, pred["mean"], rtol=1e-8, atol=1e-9)
            np.testing.assert_allclose(std, pred["std"], rtol=1e-6, atol=1e-8)
    # scikit-learn kernel objects, the way gpet.py:165-178,253 builds them
    sk = pytest.importorskip("sklearn.gaussian_process.kernels")
    from gaussian_process_edge_trace_amd.sklearn_gpr import add_white
    kern = add_white(sk.ConstantKernel(2.5, "fixed") * sk.RBF(15.0, "fixed"),
                     WeightedWhiteKernel(noise_weight=w, edge_length=xq.size, noise_level=0.4))
    gp = GaussianProcessRegressor(kernel=kern, alpha=1e-6, normalize_y=True, _ctx=ctx).fit(x[:, None], y)
    fit = orc.gp_fit(x, y, w, 2.5, 15.0, "RBF", 2.5, 0.4, xq.size, jitter=1e-6)
    np.testing.assert_allclose(gp.predict(xq[:, None]), orc.gp_predict(fit, xq, want_cov=False)["mean"], rtol=1e-8)


def test_gaussian_process_regressor_optimizer(amd, ctx):
    """GaussianProcessRegressor(optimizer="fmin_l_bfgs_b", n_restarts_optimizer=4) (sklearn_gpr.py:254-295, 587-607) on the
    device: the start points are the reference's (theta0, then RandomState(seed).uniform over the log bounds, one restart at
    a time), every start reaches the minimum scipy's L-BFGS-B reaches on the same (oracle) objective, the best one is kept,
    and predictions use the optimised kernel.  scikit-learn kernel objects with bounds, and the dict form."""
    import scipy.optimize
    from gaussian_process_edge_trace_amd.sklearn_gpr import GaussianProcessRegressor, WeightedWhiteKernel, add_white
    sk = pytest.importorskip("sklearn.gaussian_process.kernels")
    rng = np.random.default_rng(3)
    x = np.sort(rng.choice(np.arange(0, 300), size=60, replace=False)).astype(float)
    y = 4 * np.sin(x / 25.0) + rng.normal(0, 0.25, x.size) + 1.0
    w = np.ones(x.size)
    w[[0, -1]] = 1e-7
    xq = np.arange(0, 300, dtype=float)
    bounds = np.array([[1e-2, 1e3], [1.0, 200.0], [1e-4, 1.0]])
    white = WeightedWhiteKernel(noise_weight=w, edge_length=xq.size, noise_level=0.3, noise_level_bounds=tuple(bounds[2]))
    kern = add_white(sk.ConstantKernel(2.0, tuple(bounds[0])) * sk.RBF(20.0, tuple(bounds[1])), white)
    gp = GaussianProcessRegressor(kernel=kern, alpha=1e-6, optimizer="fmin_l_bfgs_b", n_restarts_optimizer=4, normalize_y=False,
                                  random_state=11, _ctx=ctx).fit(x[:, None], y)
    # the reference's start points and scipy on the oracle's objective (standardised y: normalize_y=False standardises)
    yt = (y - y.mean()) / y.std()
    lb = np.log(bounds)
    starts = [np.log([2.0, 20.0, 0.3])]
    r = np.random.RandomState(11)
    for _ in range(4):
        starts.append(r.uniform(lb[:, 0], lb[:, 1]))

    def obj(th):
        lml, g = orc.lml_and_grad(th, x, yt, w, "RBF", 2.5)
        return -lml, -g
    res = [scipy.optimize.minimize(obj, th0, method="L-BFGS-B", jac=True, bounds=list(map(tuple, lb))) for th0 in starts]
    best = min(res, key=lambda q: q.fun)
    np.testing.assert_allclose(-gp.log_marginal_likelihood_value_, best.fun, rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(gp.kernel_theta_, best.x, rtol=0, atol=2e-3)
    c, l, nl = np.exp(gp.kernel_theta_)
    fit = orc.gp_fit(x, y, w, c, l, "RBF", 2.5, nl, xq.size, jitter=1e-6, center=True, scale=True)
    pred = orc.gp_predict(fit, xq, want_cov=False)
    mean, std = gp.predict(xq[:, None], return_std=True)
    np.testing.assert_allclose(mean, pred["mean"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(std, pred["std"], rtol=1e-5, atol=1e-7)
    # dict kernel with explicit bounds, no restarts: the optimum from theta0 alone
    kd = dict(kernel="RBF", nu=2.5, constant=2.0, length_scale=20.0, white=white, bounds=bounds)
    gp2 = GaussianProcessRegressor(kernel=kd, alpha=1e-6, optimizer="fmin_l_bfgs_b", normalize_y=False, _ctx=ctx).fit(x[:, None], y)
    np.testing.assert_allclose(-gp2.log_marginal_likelihood_value_, res[0].fun, rtol=1e-7, atol=1e-7)
    with pytest.raises(NotImplementedError):
        GaussianProcessRegressor(kernel=kd, optimizer=lambda *a: None, _ctx=ctx)


def test_gaussian_process_regressor_restarts_over_free_dims_lml_value_and_cache(amd, ctx):
    """(1) Restarts are drawn over the NON-FIXED hyper-parameters only (the reference draws over kernel_.bounds,
    sklearn_gpr.py:283-288): with the noise level fixed a restart takes two numbers from RandomState, not three, so the
    second restart onwards still starts where the reference's does.  (2) optimizer=None: log_marginal_likelihood_value_ is
    the LML at the kernel's own hyper-parameters (sklearn_gpr.py:296-299), not None.  (3) the per-shape cache of device
    batches is bounded (two entries), refits with a few more points reuse an arena."""
    import scipy.optimize
    from gaussian_process_edge_trace_amd.sklearn_gpr import GaussianProcessRegressor, WeightedWhiteKernel
    rng = np.random.default_rng(5)
    x = np.sort(rng.choice(np.arange(0, 200), size=40, replace=False)).astype(float)
    y = 3 * np.cos(x / 18.0) + rng.normal(0, 0.2, x.size) - 2.0
    w = np.ones(x.size)
    xq = np.arange(0, 200, dtype=float)
    bounds = np.array([[1e-2, 1e3], [1.0, 150.0], [0.25, 0.25]])  # noise level fixed
    white = WeightedWhiteKernel(noise_weight=w, edge_length=xq.size, noise_level=0.25)
    kd = dict(kernel="RBF", nu=2.5, constant=1.5, length_scale=15.0, white=white, bounds=bounds)
    gp = GaussianProcessRegressor(kernel=kd, alpha=1e-6, optimizer="fmin_l_bfgs_b", n_restarts_optimizer=3, normalize_y=False,
                                  random_state=7, _ctx=ctx).fit(x[:, None], y)
    yt = (y - y.mean()) / y.std()
    lb = np.log(bounds)
    th0 = np.log([1.5, 15.0, 0.25])
    r = np.random.RandomState(7)
    starts = [th0]
    for _ in range(3):
        th = th0.copy()
        th[:2] = r.uniform(lb[:2, 0], lb[:2, 1])
        starts.append(th)

    def obj(th):
        lml, g = orc.lml_and_grad(th, x, yt, w, "RBF", 2.5)
        return -lml, -g
    res = [scipy.optimize.minimize(obj, s0, method="L-BFGS-B", jac=True, bounds=list(map(tuple, lb))) for s0 in starts]
    best = min(res, key=lambda q: q.fun)
    np.testing.assert_allclose(-gp.log_marginal_likelihood_value_, best.fun, rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(gp.kernel_theta_, best.x, rtol=0, atol=2e-3)
    assert gp.kernel_theta_[2] == th0[2]
    # (2)
    gp0 = GaussianProcessRegressor(kernel=kd, alpha=1e-6, optimizer=None, normalize_y=False, _ctx=ctx).fit(x[:, None], y)
    lml0, _ = orc.lml_and_grad(th0, x, yt, w, "RBF", 2.5)
    np.testing.assert_allclose(gp0.log_marginal_likelihood_value_, lml0, rtol=1e-9)
    # (3)
    gp3 = GaussianProcessRegressor(kernel=dict(kernel="RBF", nu=2.5, constant=1.5, length_scale=15.0), alpha=1e-2,
                                   normalize_y=True, _ctx=ctx)
    for n in (40, 38, 36, 33):  # capacities round up to 64: one arena
        gp3.fit(x[:n, None], y[:n]).predict(xq[:, None])
    assert len(gp3._batches) == 1
    for Lq in (200, 180, 160, 140):
        m_ = gp3.predict(np.arange(Lq, dtype=float)[:, None])
        assert m_.shape == (Lq,) and len(gp3._batches) <= 2
    for g_ in (gp, gp0, gp3):
        g_.close()


def test_gaussian_process_regressor_mirror_many_points(amd, ctx):
    """The GPR mirror beyond the LDS-resident sizes: 300 training points (K and the factor in HBM)."""
    from gaussian_process_edge_trace_amd.sklearn_gpr import GaussianProcessRegressor, WeightedWhiteKernel
    rng = np.random.default_rng(1)
    x = np.sort(rng.choice(np.arange(0, 1000), size=300, replace=False)).astype(float)
    y = 5 * np.sin(x / 60.0) + rng.normal(0, 0.3, x.size) + 3.0
    w = np.ones(x.size)
    xq = np.arange(0, 1000, dtype=float)
    kern = dict(kernel="RBF", nu=2.5, constant=2.5, length_scale=40.0,
                white=WeightedWhiteKernel(noise_weight=w, edge_length=xq.size, noise_level=0.4))
    gp = GaussianProcessRegressor(kernel=kern, alpha=1e-6, normalize_y=False, _ctx=ctx).fit(x[:, None], y)
    mean, std = gp.predict(xq[:, None], return_std=True)
    fit = orc.gp_fit(x, y, w, 2.5, 40.0, "RBF", 2.5, 0.4, xq.size, jitter=1e-6, center=True, scale=True)
    pred = orc.gp_predict(fit, xq, want_cov=False)
    np.testing.assert_allclose(mean, pred["mean"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(std, pred["std"], rtol=1e-5, atol=1e-7)


def test_topk_sort_and_rank_counting_agree_with_stable_argsort(amd, ctx, monkeypatch):
    """Top-k of the curve costs (gpet.py:443): the bitonic sort of (cost, index) in LDS and the rank-counting kernel give
    the indices and costs of numpy's stable argsort -- on costs with heavy ties, negative zeros and infinities, for sample
    counts that are and are not powers of two."""
    L = amd._lib
    rng = np.random.default_rng(7)
    for S in (64, 100, 1000):
        img, truth = orc.synth_sinusoid_image(64, 1)
        grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
        init = truth[[0, -1], :][:, [1, 0]]
        tr = amd.GP_Edge_Tracing(init, grad, kernel_options={'kernel': 'RBF', 'sigma_f': 10, 'length_scale': 8}, noise_y=1,
                                 N_samples=S, score_thresh=1, delta_x=5, keep_ratio=0.25, pixel_thresh=3, seed=1,
                                 fix_endpoints=True, _ctx=ctx)
        b = tr._batch
        b.fit_predict(want_cov=True); b.factor(); b.normals([3]); b.sample(); b.score()
        n_keep = b.info()["n_keep"]
        costs = np.round(rng.uniform(0.0, 4.0, size=S), 1)          # ~40 distinct values: ties everywhere
        costs[rng.integers(0, S, size=3)] = np.inf
        costs[rng.integers(0, S, size=3)] = 0.0
        costs[rng.integers(0, S, size=2)] = -0.0
        want = np.argsort(costs, kind="stable")[:n_keep]
        for env in (0, 1):
            L.set_option("topk_rank", env)
            b.write(L.BUF_COSTS, costs)
            b.profile_stage(141, 1)
            got_idx = b.read(L.BUF_BEST_IDX)[:n_keep]
            got_cost = b.read(L.BUF_BEST_COSTS)[:n_keep]
            assert np.array_equal(got_idx, want), (S, env)
            assert np.array_equal(got_cost, costs[want])
        L.set_option("topk_rank", 0)


def test_philox_normals_equal_the_oracle_generator(amd, ctx, golden):
    """gpet_batch_set_rng(1): the device's Philox4x32-10 + Box-Muller normals against oracle.philox_standard_normal (the
    integer part is exact; log / sincospi differ from numpy's by ulps), through the stage API, for the stored columns;
    switching back gives numpy's RandomState stream again."""
    L = amd._lib
    g = golden("stage_rbf64")
    tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **CTOR["stage_rbf64"], _ctx=ctx, rng="philox")
    b = tr._batch
    b.set_obs(0, g["in_obs"])
    b.fit_predict(want_cov=True)
    b.factor()
    b.normals([77])
    Z = b.read(L.BUF_NORMALS)
    want = orc.philox_standard_normal(77, Z.shape[0], Z.shape[1])
    np.testing.assert_allclose(Z, want, rtol=0, atol=1e-13)
    b.set_rng("mt19937")
    b.normals([77])
    Zm = b.read(L.BUF_NORMALS)
    ref = orc.legacy_standard_normal(77, Z.shape[0] * tr.x_grid.size).reshape(Z.shape[0], -1)[:, :Z.shape[1]]
    np.testing.assert_allclose(Zm, ref, rtol=0, atol=1e-14)


@pytest.mark.parametrize("name", ["trace_rbf64", "trace_mat128", "trace_rbf500"])
def test_full_trace_philox_mode_vs_oracle(amd, ctx, golden, name):
    """rng="philox" end to end: the device loop with the counter-based generator against the oracle fed with the same
    normals (oracle.trace(rng="philox")): observation sets per iteration, iteration count and edge trace bit-exact --
    the same tier as the default mode, on the other generator (whose numbers are NOT the reference's)."""
    from tests.test_oracle_vs_golden import TRACES
    g = golden(name)
    stage = TRACES[name]
    grad = golden(stage)["ref_grad"]
    kw = dict(CTOR[stage])
    rec = []
    et_o, ci_o, info = orc.trace(g["in_init"], grad, record=rec, sign_convention="harmonic", rng="philox", **kw)
    tr = amd.GP_Edge_Tracing(g["in_init"], grad, **kw, rng="philox", _ctx=ctx)
    et, (all_samples, all_obs, curves) = tr(return_lines=True)
    assert tr._n_iter == info["n_iter"]
    for i, r in enumerate(rec):
        assert np.array_equal(all_obs[i + 1], r["obs_out"]), "iteration %d" % i
    assert np.array_equal(et, et_o)


def test_jacobi_rotation_log_form_equals_lds_form(amd, ctx, golden):
    """Batches of up to 16 edges log the rotations of the LDS Jacobi and form the eigenvectors in a second kernel
    (k_jacobi_wpass: one wave per row of W in registers, seats moved by DPP shifts) instead of accumulating them in LDS
    inside the rounds: the same rotations applied with the same arithmetic -- eigenvalues, factor rows and the
    structured loop's scaled eigenvectors are IDENTICAL, on the generic path (stage API) and on the loop path."""
    L = amd._lib
    for name in ("stage_rbf64", "stage_rbf500"):
        g = golden(name)
        out = {}
        old = L.set_option("jacobi_logw", 1)
        try:
            for mode in (0, 1):
                L.set_option("jacobi_logw", mode)
                tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **CTOR[name], _ctx=ctx)
                b = tr._batch
                b.set_obs(0, g["in_obs"])
                b.fit_predict(want_cov=True)
                b.factor()
                gen = (b.read(L.BUF_FACTOR).copy(), b.read(L.BUF_EIGVALS).copy())
                b.set_obs(0, g["in_obs"])
                for stage in (120, 121, 122, 123):
                    b.profile_stage(stage, 1)
                out[mode] = gen + (b.read(L.BUF_FACTOR).copy(), b.read(L.BUF_EIGVALS).copy())
                tr2 = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **CTOR[name], _ctx=ctx)
                out[mode] += (tr2(),)
        finally:
            L.set_option("jacobi_logw", old)
        for k in range(5):
            assert np.array_equal(out[0][k], out[1][k]), (name, k)
