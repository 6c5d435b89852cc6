"""GPU parity of pixel selection (f1) and of the whole device-resident loop (a8) vs the
CPU oracle and the reference-generated fixtures."""
import numpy as np
import pytest

from oracle import gpet_oracle as orc
from tests.test_oracle_vs_golden import CTOR, STAGES, TRACES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import gaussian_process_edge_trace_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def ctx(amd):
    return amd._lib.Context(0)


@pytest.mark.parametrize("name", STAGES)
def test_gradient_kde(amd, ctx, golden, name):
    """ctor gradient KDE (gpet.py:127): float32-normalised map vs the reference (stand-in KDE)."""
    g = golden(name)
    tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **CTOR[name], _ctx=ctx)
    gk = tr._batch.read(amd._lib.BUF_GRAD_KDE)
    ref = g["ref_grad_kde"]
    # direct 9x9 convolution vs the oracle's scipy.signal.convolve: a few float32 ulps
    np.testing.assert_allclose(gk, ref, rtol=0, atol=4e-7)
    assert np.mean(gk != ref) < 0.02


@pytest.mark.parametrize("name", STAGES)
def test_curve_kde_and_pixel_selection(amd, ctx, golden, name):
    """Reference samples in -> device scoring, KDE, thresholding, binning, argmax ->
    the reference's new observation set, bit-exact, and its decayed score threshold."""
    g = golden(name)
    L = amd._lib
    ref = g["ref_samples_head"]
    tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **CTOR[name], _ctx=ctx)
    b = tr._batch
    b.set_obs(0, g["in_obs"])
    b.write(L.BUF_GRAD_KDE, g["ref_grad_kde"])  # pin the gradient KDE to the reference's
    if ref.shape[1] != int(g["ref_scalars"][2]):
        # the fixture keeps only a head of the 1000 samples (500^2 README shape): the reference's kept curves and
        # their costs go in through the get_best_pixels(curves, costs, pre_fobs) seam instead (gpet.py:622-662)
        ys = g["ref_best_curves_y"]
        curves = np.stack([np.repeat(tr.x_grid[:, None].astype(np.float64), ys.shape[1], axis=1), ys], axis=-1)
        fobs = tr.get_best_pixels(curves, g["ref_best_costs"], g["in_obs"][:, [1, 0]])
        np.testing.assert_allclose(b.read(L.BUF_KDE), g["ref_kde_arr"], rtol=0, atol=4e-7)
        assert np.array_equal(fobs, g["ref_fobs"])
        assert b.scalars().score_thresh == float(g["ref_score_thresh_out"])
        return
    b.write(L.BUF_SAMPLES, np.ascontiguousarray(ref.T))
    b.score()
    b.select_pixels()
    kde = b.read(L.BUF_KDE)
    np.testing.assert_allclose(kde, g["ref_kde_arr"], rtol=0, atol=4e-7)
    fobs = b.read(L.BUF_OBS)
    assert np.array_equal(fobs, g["ref_fobs"])
    s = b.scalars()
    assert s.score_thresh == float(g["ref_score_thresh_out"])
    assert s.iter == 1


def test_pixel_selection_500_from_reference_kde(amd, ctx, golden):
    """500x500: inject the reference's KDE array and previous observations; the device's
    scoring / threshold decay / per-bin argmax must reproduce the reference's 55 pixels."""
    g = golden("stage_rbf500")
    L = amd._lib
    tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **CTOR["stage_rbf500"], _ctx=ctx)
    b = tr._batch
    b.set_obs(0, g["in_obs"])
    b.write(L.BUF_GRAD_KDE, g["ref_grad_kde"])
    b.write(L.BUF_KDE, g["ref_kde_arr"])
    # run only the pixel stage on the injected KDE
    rc = b.lib.gpet_select_pixels_only(b.h)
    b.ctx.check(rc)
    assert np.array_equal(b.read(L.BUF_OBS), g["ref_fobs"])
    assert b.scalars().score_thresh == float(g["ref_score_thresh_out"])


@pytest.mark.parametrize("name", ["trace_rbf64", "trace_rbf65", "trace_mat128", "trace_mat35_96", "trace_rbf500"])
def test_full_trace_vs_oracle(amd, ctx, golden, name):
    """Whole trace on the device vs the oracle run with the library's eigenvector sign
    convention: observation sets per iteration, iteration count and edge trace bit-exact."""
    g = golden(name)
    stage = TRACES[name]
    grad = golden(stage)["ref_grad"]
    kw = dict(CTOR[stage])
    rec = []
    et_o, ci_o, info = orc.trace(g["in_init"], grad, record=rec, sign_convention="harmonic", **kw)
    tr = amd.GP_Edge_Tracing(g["in_init"], grad, **kw, _ctx=ctx)
    et, (all_samples, all_obs, curves) = tr(return_lines=True)
    assert tr._n_iter == info["n_iter"]
    for i, r in enumerate(rec):
        assert np.array_equal(all_obs[i + 1], r["obs_out"]), "iteration %d" % i
    assert np.array_equal(et, et_o)
    assert et.shape == g["ref_edge_trace"].shape and et.dtype == g["ref_edge_trace"].dtype
    kw["return_std"] = True
    tr2 = amd.GP_Edge_Tracing(g["in_init"], grad, **kw, _ctx=ctx)
    et2, ci = tr2()
    assert np.array_equal(et2, et)
    np.testing.assert_allclose(ci[0], ci_o[0], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ci[1], ci_o[1], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", ["trace_rbf64", "trace_rbf65", "trace_mat128", "trace_rbf500"])
def test_full_trace_f32_samples_vs_oracle(amd, ctx, golden, name):
    """Opt-in sample_dtype="f32" (gpet_batch_set_sample_dtype; BASELINE config 2's "fp32 posterior samples"): the GEMM
    rounds every sample to f32 when it stores it, scorer / KDE / pixel kernels widen it -- all arithmetic f64.  Against
    the oracle with `y_samples.astype(float32)` after sample_y (sample_dtype="f32"): observation sets per iteration,
    iteration count and edge trace bit-exact; the samples the stage API returns are f32-representable; and the
    trace is a different one than the f64 trace only where rounding to f32 moves a decision (printed)."""
    g = golden(name)
    stage = TRACES[name]
    grad = golden(stage)["ref_grad"]
    kw = dict(CTOR[stage])
    rec = []
    et_o, ci_o, info = orc.trace(g["in_init"], grad, record=rec, sign_convention="harmonic", sample_dtype="f32", **kw)
    tr = amd.GP_Edge_Tracing(g["in_init"], grad, **kw, sample_dtype="f32", _ctx=ctx)
    et, (all_samples, all_obs, curves) = tr(return_lines=True)
    assert tr._n_iter == info["n_iter"]
    for i, r in enumerate(rec):
        assert np.array_equal(all_obs[i + 1], r["obs_out"]), "iteration %d" % i
    assert np.array_equal(et, et_o)
    # the stage API on the same object: samples come back as f64 values that are exactly f32
    b = tr._batch
    b.set_obs(0, g["ref_obs_01"] if "ref_obs_01" in g else all_obs[1])
    b.fit_predict(want_cov=True)
    b.factor()
    b.normals([kw.get("seed", 42) + 1])
    b.sample()
    Y = b.read(amd._lib.BUF_SAMPLES)
    assert Y.dtype == np.float64 and np.array_equal(Y, Y.astype(np.float32).astype(np.float64))
    tr64 = amd.GP_Edge_Tracing(g["in_init"], grad, **kw, _ctx=ctx)
    et64 = tr64()
    print("%s: f32-sample trace %s the f64-sample trace (%d vs %d iterations)"
          % (name, "equals" if np.array_equal(et, et64) else "differs from", tr._n_iter, tr64._n_iter))


def test_trace_quality_band(amd, ctx):
    """Trace quality vs ground truth on a synthetic image where the reference algorithm itself is
    stable across random draws (image seed 3: oracle MSE 39-52 over sign conventions and seeds;
    image seed 1 swings between 692 and 8443 in the reference too, so it is no quality gate)."""
    img, truth = orc.synth_sinusoid_image(500, 3)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    tr = amd.GP_Edge_Tracing(init, grad, **CTOR["stage_rbf500"], _ctx=ctx)
    et = tr()
    assert np.array_equal(et[:, 1], np.arange(500))
    assert amd.gpet_utils.trace_MSE(et, truth) < 150.0
    assert amd.gpet_utils.trace_dicecoef(et, truth) > 0.97
    assert 8 <= tr._n_iter <= 40


def test_lml_kernel_vs_oracle(amd, ctx, golden):
    """f2: batched -log marginal likelihood + gradient (sklearn_gpr.py:512-585) vs the oracle."""
    from tests import final_fit_inputs as ff
    for name, stage in [("trace_rbf500", "stage_rbf500"), ("trace_mat128", "stage_mat128")]:
        g = golden(name)
        grad = golden(stage)["ref_grad"]
        kw = CTOR[stage]
        tr = amd.GP_Edge_Tracing(g["in_init"], grad, **kw, _ctx=ctx)
        last = max(int(k[8:]) for k in g if k.startswith("ref_obs_"))
        obs = g["ref_obs_%02d" % last]
        pr = ff.prepare(tr.init, obs, tr.x_grid, tr.fix_endpoints)
        b = tr._batch
        b.final_set_training(0, pr["xs"], pr["yt"], pr["w"])
        rng = np.random.default_rng(0)
        th = ff.BOUNDS[:, 0] + (ff.BOUNDS[:, 1] - ff.BOUNDS[:, 0]) * rng.uniform(size=(40, 3))
        th[:, 2] = np.log(rng.uniform(1e-4, 1.0, size=40))  # keep most of them positive definite
        th[0] = np.log([5.0, 5.0, 1.0])
        f, gr = b.lml_batch(np.zeros(40, dtype=np.int32), th)
        for i in range(40):
            lml, g_o = orc.lml_and_grad(th[i], pr["xs"], pr["yt"], pr["w"], tr.kernel_type, tr.kernel_nu)
            if not np.isfinite(lml):
                assert np.isinf(f[i]) and f[i] > 0
                continue
            np.testing.assert_allclose(f[i], -lml, rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(gr[i], -g_o, rtol=1e-6, atol=1e-6 * (1 + np.abs(g_o).max()))


def test_final_fit_matches_reference_theta(amd, ctx, golden):
    """The converged fit driven in lock step with the device objective lands on the reference's
    optimum (theta and CI from the reference run, trace_* fixtures, given its observations)."""
    from gaussian_process_edge_trace_amd.gpet import device_final_fits
    for name, stage in [("trace_rbf64", "stage_rbf64"), ("trace_rbf65", "stage_rbf65"),
                        ("trace_mat128", "stage_mat128"), ("trace_rbf500", "stage_rbf500"),
                        ("trace_mat35_96", "stage_mat35_96")]:
        g = golden(name)
        grad = golden(stage)["ref_grad"]
        kw = dict(CTOR[stage])
        tr = amd.GP_Edge_Tracing(g["in_init"], grad, **kw, _ctx=ctx)
        n_iter = int(g["ref_n_iter"])
        obs = g["ref_obs_%02d" % n_iter]
        fits, rounds = device_final_fits(tr._batch, [dict(tr._p, seed=tr.seed)], [obs], [n_iter])
        mean, std, theta = fits[0]
        if name == "trace_mat35_96":
            # general nu: sklearn's length-scale gradient is a forward difference with step 1e-10 (noise ~1e-6 of the
            # gradient), the device's is analytic: the optimiser paths part at that level
            np.testing.assert_allclose(theta[:2], g["ref_final_theta"][:2], rtol=0, atol=5e-3)
            np.testing.assert_allclose(mean, g["ref_final_mean"], rtol=0, atol=5e-3)
            assert np.array_equal(np.rint(mean).astype(int), g["ref_edge_trace"][:, 0])
            continue
        # log c and log l to 1e-4; the noise level enters K as nl * w + 1e-6, so it is compared in linear
        # space (the objective is flat in log nl once nl << 1e-6 and the optimiser's stopping point there
        # is decided by rounding noise in the gradient)
        np.testing.assert_allclose(theta[:2], g["ref_final_theta"][:2], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(np.exp(theta[2]), np.exp(g["ref_final_theta"][2]), rtol=1e-3, atol=1e-9)
        np.testing.assert_allclose(mean, g["ref_final_mean"], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(mean - 1.96 * std, g["ref_ci_lower"], rtol=1e-5, atol=1e-4)
        assert np.array_equal(np.rint(mean).astype(int), g["ref_edge_trace"][:, 0])


def test_device_training_sets_and_start_points_equal_numpy(amd, ctx, golden):
    """gpet_final_fit_all builds its inputs on the device: sorted training set standardised twice (gpet.py:235-238,
    sklearn_gpr.py:229-234) with numpy's pairwise summation order, and theta0 + 12 restarts from MT19937
    (sklearn_gpr.py:283-288).  Bit-identical to the NumPy restatement (= the reference's own arithmetic)."""
    from tests import final_fit_inputs as ff
    L = amd._lib
    for name, stage in [("trace_rbf500", "stage_rbf500"), ("trace_mat128", "stage_mat128"), ("trace_rbf65", "stage_rbf65")]:
        g = golden(name)
        tr = amd.GP_Edge_Tracing(g["in_init"], golden(stage)["ref_grad"], **CTOR[stage], _ctx=ctx)
        n_iter = int(g["ref_n_iter"])
        obs = g["ref_obs_%02d" % n_iter]
        b = tr._batch
        b.set_obs(0, obs)
        seed = tr.seed + n_iter
        b.final_fit_all([seed])
        pr = ff.prepare(tr.init, obs, tr.x_grid, tr.fix_endpoints)
        n = pr["xs"].shape[0]
        train = b.read(L.BUF_FIN_TRAIN)
        assert np.array_equal(train[0, :n], pr["xs"]) and np.array_equal(train[1, :n], pr["yt"])
        assert np.array_equal(train[2, :n], pr["w"])
        par = b.read(L.BUF_FIN_PAR)
        assert np.array_equal(par[3:9], [pr["X_m"], pr["X_s"], pr["y_m"], pr["y_s"], pr["m2"], pr["s2"]])
        assert np.array_equal(b.read(L.BUF_FIN_STARTS), np.asarray(ff.start_points(tr.noise_y, seed)))


def test_device_lbfgsb_against_scipy_on_the_same_objective(amd, ctx, golden):
    """The device's L-BFGS-B state machines against scipy.optimize.minimize(method="L-BFGS-B") (what the reference
    calls, sklearn_gpr.py:589) driving the SAME device objective from the same 13 start points: per restart the same
    minimum to 1e-7 relative (flat noise directions aside, theta to 1e-3), and the same best restart."""
    import scipy.optimize
    from tests import final_fit_inputs as ff
    L = amd._lib
    for name, stage in [("trace_rbf500", "stage_rbf500"), ("trace_mat128", "stage_mat128")]:
        g = golden(name)
        tr = amd.GP_Edge_Tracing(g["in_init"], golden(stage)["ref_grad"], **CTOR[stage], _ctx=ctx)
        n_iter = int(g["ref_n_iter"])
        b = tr._batch
        b.set_obs(0, g["ref_obs_%02d" % n_iter])
        mean, std, theta, fmin, rounds = b.final_fit_all([tr.seed + n_iter])
        assert 5 < rounds < 400
        starts = b.read(L.BUF_FIN_STARTS)

        def obj(th):
            f, gr = b.lml_batch(np.zeros(1, dtype=np.int32), np.asarray(th).reshape(1, 3))
            return float(f[0]), gr[0].copy()
        res = [scipy.optimize.minimize(obj, th0, method="L-BFGS-B", jac=True, bounds=list(map(tuple, ff.BOUNDS))) for th0 in starts]
        fs = np.array([r.fun for r in res])
        best = int(np.argmin(fs))
        np.testing.assert_allclose(fmin[0], fs[best], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(theta[0][:2], res[best].x[:2], rtol=0, atol=1e-3)
        np.testing.assert_allclose(theta[0][:2], g["ref_final_theta"][:2], rtol=1e-4, atol=1e-5)


def test_final_fit_one_workgroup_per_problem_equals_rounds(amd, ctx, golden):
    """gpet_set_option("fit_persistent"): the converged fit as one workgroup per (edge, restart) problem (k_lml16_fit:
    objective and L-BFGS-B state machine alternate inside the workgroup) against the lock-step rounds of launches --
    the same objective source and the same state-machine code; the two kernels the objective is compiled into round
    differently in the last bit here and there (README edge: theta 2e-13, mean 2e-12 apart; 65-px edge, whose noise level
    ends in the flat part of the objective: mean 3e-7), so: log amplitude and length scale to 1e-6, noise level compared
    in linear space, best objective value to 1e-9 relative, mean / std to 1e-5, evaluation count of the longest
    restart within 3 of the number of rounds."""
    L = amd._lib
    for name, stage in [("trace_rbf500", "stage_rbf500"), ("trace_mat128", "stage_mat128"), ("trace_rbf65", "stage_rbf65")]:
        g = golden(name)
        tr = amd.GP_Edge_Tracing(g["in_init"], golden(stage)["ref_grad"], **CTOR[stage], _ctx=ctx)
        n_iter = int(g["ref_n_iter"])
        b = tr._batch
        b.set_obs(0, g["ref_obs_%02d" % n_iter])
        out = {}
        old = L.set_option("fit_persistent", 0)
        try:
            for mode in (0, 1):
                L.set_option("fit_persistent", mode)
                out[mode] = b.final_fit_all([tr.seed + n_iter])
        finally:
            L.set_option("fit_persistent", -1 if old == 2 else old)
        t0, t1 = np.asarray(out[0][2]), np.asarray(out[1][2])
        np.testing.assert_allclose(t1[:, :2], t0[:, :2], rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.exp(t1[:, 2]), np.exp(t0[:, 2]), rtol=1e-3, atol=1e-9)  # (flat direction: as above)
        np.testing.assert_allclose(out[1][3], out[0][3], rtol=1e-9, atol=0)
        np.testing.assert_allclose(out[1][0], out[0][0], rtol=0, atol=1e-5)
        np.testing.assert_allclose(out[1][1], out[0][1], rtol=0, atol=1e-5)
        assert abs(out[0][4] - out[1][4]) <= 3, (name, out[0][4], out[1][4])


def test_structured_loop_path_equals_generic(amd, ctx, golden):
    """The loop's prior-eigenbasis path (H = c Lam - U^T U in the eigenbasis of the grid's Toeplitz
    correlation matrix) must produce the factor, mean and samples of the generic path
    (covariance -> pivoted Cholesky -> Gram -> Jacobi) for the same observations."""
    L = amd._lib
    g = golden("stage_rbf500")
    tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **CTOR["stage_rbf500"], _ctx=ctx)
    b = tr._batch
    info = b.info()
    assert info["structured"] == 1 and 40 < info["r0"] < 96
    # generic path through the per-stage API
    b.set_obs(0, g["in_obs"])
    b.fit_predict(want_cov=True)
    b.factor()
    A_gen = b.read(L.BUF_FACTOR)
    ev_gen = b.read(L.BUF_EIGVALS)
    mean_gen = b.read(L.BUF_MEAN)
    cov = b.read(L.BUF_COV)
    # structured path: one loop iteration on the same observation set (stages 120-123)
    b.set_obs(0, g["in_obs"])
    for stage in (120, 121, 122, 123):
        b.profile_stage(stage, 1)
    A_str = b.read(L.BUF_FACTOR)
    ev_str = b.read(L.BUF_EIGVALS)
    np.testing.assert_allclose(b.read(L.BUF_MEAN), mean_gen, rtol=1e-10)
    scale = ev_gen[0]
    np.testing.assert_allclose(A_str.T @ A_str, cov, rtol=0, atol=1e-9 * scale)
    k = min(A_gen.shape[0], A_str.shape[0])
    y_std = b.scalars().y_std
    np.testing.assert_allclose(ev_str[:k] * y_std ** 2, ev_gen[:k], rtol=1e-6, atol=1e-11 * scale)
    sig = ev_gen[:k] > 1e-9 * scale
    for i in np.nonzero(sig)[0]:
        np.testing.assert_allclose(A_str[i], A_gen[i], rtol=0, atol=2e-5 * np.sqrt(ev_gen[i]) + 1e-9 * np.sqrt(scale))
