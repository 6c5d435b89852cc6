"""Round 6: the device held against the UNMODIFIED reference directly -- no oracle in between.

(1) Replay of the reference's own traces.  LAPACK's singular-vector signs are the one thing of a reference run the library
cannot compute (they depend on the BLAS build and its thread count), so the fixtures carry them: per
iteration one bit per row of the reference's factor sqrt(s) v (tests/golden/make_fixtures.py hooks numpy.linalg.svd).  The
test drives the per-stage C ABI iteration by iteration (fit -> factor -> normals -> flip the N(0, 1) columns whose reference
row points the other way -> samples -> scores -> pixel selection) and must reproduce the reference's observation set after
EVERY iteration and its final edge trace with array_equal (gpet.py:839-861, sklearn_gpr.py:464).
(2) BASELINE config 3's shape and config 5's frame from the reference (stage_rbf2048_n1500, stage_mat1024) at full size.
(3) The README's demo on the reference's own image (tests/golden/readme_image.npz: the unmodified construct_test_img under
scikit-image 0.18.3), called as the README writes it."""
import numpy as np
import pytest

from oracle import gpet_oracle as orc
from tests.test_oracle_vs_golden import CTOR, CTOR_BIG, README_LITERAL, TRACES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import gaussian_process_edge_trace_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def ctx(amd):
    return amd._lib.Context(0)


def _replay(amd, tr, g, n_iter, seed, slack=0):
    """the loop of gpet.py:829-870 through the stage entry points, the reference's SVD signs applied to the normals.
    slack = 0: free-running, every observation set must EQUAL the reference's.  slack > 0 (covariances whose trailing singular
    values lie below LAPACK's resolution): every iteration starts from the REFERENCE's observation set and threshold, and the
    device's new set may differ from the reference's next one in at most `slack` pixels; returns the worst difference too."""
    L = amd._lib
    b = tr._batch
    N = tr.edge_length
    b.set_obs(0, g["ref_obs_00"].reshape(-1, 2).astype(np.int64))
    worst = 0
    for i in range(n_iter):
        if slack and i > 0:
            thresh = b.scalars().score_thresh
            b.set_obs(0, g["ref_obs_%02d" % i].reshape(-1, 2).astype(np.int64))
            sc = b.scalars()
            sc.score_thresh = thresh  # (the threshold persists and only decays, Q7)
            b.write_scalars(sc)
        assert not b.scalars().done, "iteration %d: the device thinks the trace has finished" % i
        b.fit_predict(want_cov=True)
        b.factor()
        k = b.read(L.BUF_FACTOR).shape[0]
        up = np.unpackbits(g["ref_svd_sign_bits_%02d" % i])[:N].astype(bool)  # row k of the reference's factor has harmonic sum >= 0
        b.normals([seed + i + 1])  # gpet.py:839
        Z = b.read(L.BUF_NORMALS)
        kk = min(k, Z.shape[1])
        Z[:, :kk] *= np.where(up[:kk], 1.0, -1.0)[None, :]
        b.write(L.BUF_NORMALS, Z)
        b.sample()
        b.score()
        b.select_pixels()
        obs = b.read(L.BUF_OBS)
        want = g["ref_obs_%02d" % (i + 1)]
        if slack:
            diff = len(set(map(tuple, obs.tolist())) ^ set(map(tuple, want.reshape(-1, 2).tolist())))
            worst = max(worst, diff)
            assert diff <= slack, "observation set after iteration %d differs from the reference's in %d pixels" % (i, diff)
        else:
            assert obs.shape == want.shape and np.array_equal(obs, want), "observation set after iteration %d" % i
    if slack:
        b.set_obs(0, g["ref_obs_%02d" % n_iter].reshape(-1, 2).astype(np.int64))
        tr._replay_worst = worst
    assert b.scalars().done
    from gaussian_process_edge_trace_amd.gpet import device_final_fits
    fits, _ = device_final_fits(b, [dict(tr._p, seed=seed)], None, [n_iter])  # gpet.py:874: seed + N_iter
    mean, std, theta = fits[0]
    et = np.rint(np.stack([mean, tr.x_grid.astype(np.float64)], axis=1)).astype(int)  # gpet.py:885-886 (yx)
    return et, (mean - 1.96 * std, mean + 1.96 * std), theta


@pytest.mark.parametrize("name", ["trace_rbf64", "trace_rbf65", "trace_mat128", "trace_mat35_96", "trace_rbf500"])
def test_replay_of_the_reference_trace_with_its_svd_signs(amd, ctx, golden, name):
    g = golden(name)
    kw = dict(CTOR[TRACES[name]])
    grad = g["ref_grad"] if "ref_grad" in g else golden("stage_rbf500")["ref_grad"]
    tr = amd.GP_Edge_Tracing(g["in_init"], grad, **kw, _ctx=ctx)
    n_iter = int(g["ref_n_iter"])
    et, ci, theta = _replay(amd, tr, g, n_iter, kw["seed"])
    assert np.array_equal(et, g["ref_edge_trace"])
    np.testing.assert_allclose(ci[0], g["ref_ci_lower"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ci[1], g["ref_ci_upper"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(theta, g["ref_final_theta"], rtol=2e-3 if "mat35" in name else 1e-4, atol=1e-6)


def _unpack_replay(g, t):
    """trace t of a replay_* fixture in the layout _replay reads (ref_obs_NN, ref_svd_sign_bits_NN)"""
    counts = g["ref_obs_counts_%02d" % t].astype(int)
    allobs = g["ref_obs_all_%02d" % t].astype(np.int64)
    out, o = {}, 0
    for i, c in enumerate(counts):
        out["ref_obs_%02d" % i] = allobs[o:o + c]
        o += c
    for i, b_ in enumerate(g["ref_svd_sign_bits_%02d" % t]):
        out["ref_svd_sign_bits_%02d" % i] = b_
    return out


@pytest.mark.parametrize("name,kernel_options", [("replay_rbf500", {'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}),
                                                 ("replay_default500", (1, 3, 3))])
def test_replay_of_more_reference_traces(amd, ctx, golden, name, kernel_options):
    """Twelve further traces of the unmodified reference on the README configuration (two images x six seeds 997 apart) and
    two with the reference's DEFAULT kernel (kernel_options = (1, 3, 3): Matern-5/2, l = 250 -- full-rank covariances, the
    any-rank factor at the README's size), each reduced to its observation sets, SVD sign bits and final trace: every one must
    replay with array_equal.  The default kernel's covariances (length scale 250 px on 500 columns) have hundreds of singular values
    below 1e-13 of the largest: LAPACK's SVD of the COVARIANCE resolves those directions to absolute accuracy only, so the reference's
    own samples carry ~1e-3 px of rounding noise that no other solver reproduces (the device's one-sided Jacobi on the Cholesky rows
    is accurate RELATIVE to each singular value; tools/tolsweep_warm.py: 5e-3 px apart at a trace's first iteration, 1e-7 px once
    observations have cut the tail) -- a near-tie of the pixel selection then falls the other way now and then (free-running: one
    pixel of 24 at iteration 3 of the first trace).  For that fixture every iteration therefore starts from the reference's
    observation set, the device's next set may differ from the reference's in at most two pixels, and the final fit -- which draws
    nothing -- must give the reference's trace."""
    g = golden(name)
    N = int(g["in_N"])
    slack = 2 if name == "replay_default500" else 0
    kw = dict(CTOR["stage_rbf500"], kernel_options=kernel_options)
    grads = {}
    for t in range(len(g["in_seeds"])):
        img_seed, seed = int(g["in_img_seeds"][t]), int(g["in_seeds"][t])
        if img_seed not in grads:
            img, edge = orc.synth_sinusoid_image(N, img_seed)
            grads[img_seed] = (amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx), edge[[0, -1], :][:, [1, 0]])
        grad, init = grads[img_seed]
        tr = amd.GP_Edge_Tracing(init, grad, **dict(kw, seed=seed), _ctx=ctx)
        et, _, _ = _replay(amd, tr, _unpack_replay(g, t), int(g["ref_n_iter"][t]), seed, slack=slack)
        assert np.array_equal(et, g["ref_edge_trace_%02d" % t].astype(int)), (name, t, img_seed, seed)
        if slack:
            print("%s trace %d: worst per-iteration difference from the reference's observation set: %d pixel(s)" % (name, t, tr._replay_worst))
        tr._batch.close()


def test_readme_demo_on_the_reference_image_replayed_and_as_called(amd, ctx, golden):
    """README.md:46-85 on the reference's own image.  (a) the package's generator and gradient image ARE the reference's;
    (b) the replay with the reference's SVD signs reproduces its 29 observation sets and its trace, hence its MSE / relative
    area / DICE to the digit (published for real KDEpy: 12.604 / 0.00339 / 0.9953, Figures/noisy_trace_results.png -- this
    image's reference run with the KDE stand-in: the fixture's ref_metrics); (c) the call as the README writes it, with the
    library's own sign convention: another equally valid draw -- same iteration regime and quality band."""
    U = amd.gpet_utils
    ri, g = golden("readme_image"), golden("readme_trace")
    test_img, true_edge = U.construct_test_img(size=(500, 500), amplitude=200, curvature=4, noise_level=0.05, ltype='sinusoidal',
                                               intensity=0.3, gaps=True)
    assert np.array_equal(test_img, ri["ref_img"]) and np.array_equal(true_edge, ri["ref_true_edge"])
    kernel = U.kernel_builder(size=(11, 5), unit=False)
    grad_img = U.comp_grad_img(test_img, kernel, ctx=ctx)
    assert np.array_equal(grad_img, ri["ref_grad_py39"])
    kernel_params = {'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}
    delta_x, score_thresh, N_samples, noise_y, seed, keep_ratio = 5, 1, 1000, 1, 1, 0.1
    init = true_edge[[0, -1], :][:, [1, 0]]
    obs = np.array([])
    fix_endpoints, return_std = True, True
    noisy_trace = amd.GP_Edge_Tracing(init, grad_img, kernel_params, noise_y, obs, N_samples, score_thresh,
                                      delta_x, keep_ratio, seed, return_std, fix_endpoints, _ctx=ctx)
    assert [noisy_trace.pixel_thresh, noisy_trace.seed, int(noisy_trace.return_std), int(noisy_trace.fix_endpoints),
            noisy_trace.algo_thresh] == list(g["ref_bound"])  # quirk Q6
    # (b)
    et, ci, theta = _replay(amd, noisy_trace, g, int(g["ref_n_iter"]), int(noisy_trace.seed))
    assert np.array_equal(et, g["ref_edge_trace"])
    np.testing.assert_allclose(ci[0], g["ref_ci_lower"], rtol=1e-5, atol=1e-5)
    got = [U.trace_MSE(et, true_edge), U.trace_relarea(et, true_edge), U.trace_dicecoef(et, true_edge)]
    assert got == list(g["ref_metrics"])
    print("README demo, replayed: MSE %.3f  rel. area %.5f  DICE %.4f  (reference on this image %s; published with KDEpy %s)"
          % (got[0], got[1], got[2], list(g["ref_metrics"]), list(g["published_metrics"])))
    # (c) (a fresh object: like the reference's, an instance is single-use -- its score threshold only ever decays, Q7)
    noisy_trace = amd.GP_Edge_Tracing(init, grad_img, kernel_params, noise_y, obs, N_samples, score_thresh,
                                      delta_x, keep_ratio, seed, return_std, fix_endpoints, _ctx=ctx)
    edge_pred, edge_credint = noisy_trace(False, False, False, False)
    m = [U.trace_MSE(edge_pred, true_edge), U.trace_relarea(edge_pred, true_edge), U.trace_dicecoef(edge_pred, true_edge)]
    print("README demo, as called (library's sign convention): MSE %.3f  rel. area %.5f  DICE %.4f, %d iterations"
          % (m[0], m[1], m[2], noisy_trace._n_iter))
    assert edge_pred.shape == (500, 2) and edge_credint[0].shape == (500,)
    assert abs(noisy_trace._n_iter - int(g["ref_n_iter"])) <= 6
    assert m[2] >= 0.97 and m[0] <= 400.0  # (this image is bistable between two tracks of one gap: DESIGN.md section 3 (iv))


def _big_stage(amd, ctx, g, name):
    N = int(g["ref_scalars"][8])
    img, edge = orc.synth_sinusoid_image(N, int(g["in_img_seed"]))
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    assert np.array_equal(grad[::97, ::89], g["ref_grad_probe"])  # the reference's own gradient image at the probe pixels
    tr = amd.GP_Edge_Tracing(g["in_init"], grad, obs=g["in_obs"], **CTOR_BIG[name], _ctx=ctx)
    assert [tr.x_st, tr.x_en, tr.N_samples, tr.N_keep, tr.N_subints, tr.algo_thresh, tr.delta_x, tr.pixel_thresh,
            tr.edge_length] == list(g["ref_scalars"])
    np.testing.assert_allclose(tr._batch.read(amd._lib.BUF_GRAD_KDE)[::97, ::89], g["ref_grad_kde_probe"], rtol=0, atol=4e-7)
    return tr, grad, N


@pytest.mark.parametrize("name", ["stage_rbf2048_n1500", "stage_mat1024"])
def test_config3_and_config5_stage_parity_vs_the_reference(amd, ctx, golden, name):
    """BASELINE config 3 (2048^2, 1500 training points, 4000 samples: blocked f64-MFMA fit, pivoted Cholesky over the GPU)
    and config 5's frame (1024^2 Matern-5/2 warm start: the any-rank factor) against vectors of the unmodified reference:
    T1 (alpha, mean, std, covariance), the spectrum, the factor rows up to sign, T2 (sign-aligned samples), the costs of
    the reference's own samples and its best_idxs."""
    g = golden(name)
    L = amd._lib
    tr, grad, N = _big_stage(amd, ctx, g, name)
    b = tr._batch
    b.set_obs(0, g["in_obs"])
    b.fit_predict(want_cov=True)
    s = b.scalars()
    assert s.n == g["ref_X_train"].shape[0] and np.array_equal(b.read(L.BUF_X_TRAIN), g["ref_X_train"])
    np.testing.assert_allclose(b.read(L.BUF_Y_TRAIN), g["ref_y_train"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(s.amp, float(g["ref_amp"]), rtol=1e-14)
    np.testing.assert_allclose(b.read(L.BUF_ALPHA), g["ref_alpha"], rtol=1e-5, atol=1e-7 * np.abs(g["ref_alpha"]).max())
    np.testing.assert_allclose(b.read(L.BUF_MEAN), g["ref_mean"], rtol=1e-7)
    np.testing.assert_allclose(b.read(L.BUF_STD), g["ref_std"], rtol=1e-5, atol=1e-7)  # north_star: 1e-5 relative
    cov = b.read(L.BUF_COV)
    scale = np.abs(g["ref_cov_diag"]).max()
    for got, want in ((np.diag(cov), g["ref_cov_diag"]), (cov[0], g["ref_cov_row0"]), (cov[N // 2], g["ref_cov_rowmid"])):
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-9 * scale)
    b.factor()
    A, ev = b.read(L.BUF_FACTOR), b.read(L.BUF_EIGVALS)
    sv, F = g["ref_svals"], g["ref_factor_top"]
    r = A.shape[0]
    np.testing.assert_allclose(ev[:r], sv[:r], rtol=1e-6, atol=1e-10 * sv[0])
    np.testing.assert_allclose(A.T @ A, cov, rtol=0, atol=1e-9 * sv[0])
    k = min(r, F.shape[0])
    gap = np.minimum(np.abs(np.diff(sv, prepend=np.inf)), np.abs(np.diff(sv, append=-np.inf)))
    for i in np.nonzero((sv[:k] > 1e-9 * sv[0]) & (gap[:k] > 1e-6 * sv[:k]))[0]:  # (well separated directions are unique up to sign)
        sgn = np.sign(A[i] @ F[i])
        np.testing.assert_allclose(sgn * A[i], F[i], rtol=0, atol=2e-5 * np.sqrt(sv[i]) + 1e-9 * np.sqrt(sv[0]), err_msg="row %d" % i)
    # T2: the device's rows follow the library's convention (harmonic sum >= 0); the reference's own signs are in the fixture
    up = np.unpackbits(g["ref_svd_sign_bits"])[:N].astype(bool)
    b.normals([int(g["in_gp_seed"])])
    Z = b.read(L.BUF_NORMALS)
    assert np.array_equal(Z[:8, :min(Z.shape[1], N)], g["ref_Z_head"][:, :Z.shape[1]]) or np.abs(Z[:8] - g["ref_Z_head"][:, :Z.shape[1]]).max() < 1e-14
    kk = min(r, Z.shape[1])
    Z[:, :kk] *= np.where(up[:kk], 1.0, -1.0)[None, :]
    b.write(L.BUF_NORMALS, Z)
    b.sample()
    Y = b.read(L.BUF_SAMPLES).T
    ref = g["ref_samples_head"]
    ns = ref.shape[1]
    degenerate = sv[(gap < 1e-4 * sv) & (sv > 0)]
    y_s = b.scalars().y_s
    tail = sv[r:].sum() if r < sv.shape[0] else 0.0  # (directions beyond the factor capacity: the reference draws along them too)
    atol = 1e-5 * np.abs(ref).max() + 6.0 * y_s * np.sqrt(degenerate.sum() + tail)
    np.testing.assert_allclose(Y[:, :ns], ref, rtol=1e-5, atol=atol)
    # a7 on the reference's own samples (the first `ns`; the rest are the device's sign-aligned ones)
    Yall = b.read(L.BUF_SAMPLES)
    Yall[:ns] = ref.T
    b.write(L.BUF_SAMPLES, Yall)
    b.score()
    costs = b.read(L.BUF_COSTS)
    np.testing.assert_allclose(costs[:ns], g["ref_costs"][:ns], rtol=1e-9)
    np.testing.assert_allclose(costs, g["ref_costs"], rtol=1e-4)  # (sign-aligned device samples: the same curves to ~1e-6 px)
    oc = orc.costs_batch(orc.normalise(grad, (0, 1), np.float64), tr.x_grid, Yall.T)
    assert np.array_equal(b.read(L.BUF_BEST_IDX), np.argsort(oc, kind="stable")[:tr.N_keep])
    # the reference's best_idxs: equal wherever the cut at N_keep is decided by more than the samples' own ~1e-7 noise
    ref_idx = g["ref_best_idxs"]
    srt = np.sort(g["ref_costs"])
    if srt[tr.N_keep] - srt[tr.N_keep - 1] > 1e-5 * srt[tr.N_keep]:
        assert set(b.read(L.BUF_BEST_IDX).tolist()) == set(ref_idx.tolist())
    if "ref_fobs" in g:  # config 5's frame: the pixel selection on the reference's kept curves' KDE is in test_gpu_trace; here end to end
        b.select_pixels()
        fobs = b.read(L.BUF_OBS)
        want = g["ref_fobs"]
        common = len(set(map(tuple, fobs.tolist())) & set(map(tuple, want.tolist())))
        assert fobs.shape[0] == want.shape[0] and common >= 0.9 * want.shape[0], (fobs.shape, want.shape, common)
