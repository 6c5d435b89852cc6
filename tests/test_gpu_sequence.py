"""GPU parity of image-sequence tracing (BASELINE config 5: frames chained through the ``obs`` warm start,
gp_edge_tracing/gpet.py:57-61,100,820,829; Matern-5/2) against the CPU oracle chained the same way, of the reference's
per-method seams against the fixture arrays the unmodified reference produced, and of whole traces against the
REFERENCE's own traces (tier T3, as numbers)."""
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


def make_sequence(amd, ctx, N, T, seed0=11):
    """T frames of one drifting sinusoidal edge: amplitude grows 2 % per frame, fresh noise per frame; the end points
    stay on row N//2 (the wave is zero at both ends), so one ``init`` serves every frame."""
    frames, truths = [], []
    for t in range(T):
        img, truth = orc.synth_sinusoid_image(N, seed0 + t, amplitude=int(0.4 * N * (1.0 + 0.02 * t)))
        frames.append(amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx))
        truths.append(truth)
    init = truths[0][[0, -1], :][:, [1, 0]]
    return frames, truths, init


def oracle_chain(frames, init, warm_every, seeds, kw):
    """The reference's usage restated with the oracle: one trace per frame, frame t warm-started from every
    ``warm_every``-th pixel of trace t-1 (the product's own thinning rule, host logic)."""
    from gaussian_process_edge_trace_amd.sequence import warm_start_obs
    out, iters, prev = [], [], None
    for t, grad in enumerate(frames):
        p = orc.resolve_params(init, grad, **kw)
        obs = (np.zeros((0, 2), dtype=np.int64) if prev is None else
               warm_start_obs(prev, p["x_st"], p["x_en"], warm_every, p["algo_thresh"], p["M"]))
        et, ci, info = orc.trace(init, grad, obs=obs, seed=seeds[t], sign_convention="harmonic", **kw)
        out.append(et)
        iters.append(info["n_iter"])
        prev = et
    return out, iters


@pytest.mark.parametrize("N,T,chains", [(512, 6, 2), (1024, 4, 2)])
def test_matern_sequence_bit_exact_vs_chained_oracle(amd, ctx, N, T, chains):
    """Matern-5/2 (full-rank posterior: the any-rank factor of gpet_eig.hip), T frames in `chains` chains traced as
    batches of `chains` edges: every frame's trace and iteration count equal the oracle's chained run bit for bit --
    cold first frames and warm-started later ones."""
    frames, truths, init = make_sequence(amd, ctx, N, T)
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1,
              N_samples=300, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    seeds = [3 + t for t in range(T)]
    st = amd.SequenceTracer(frames, init, n_chains=chains, warm_every=16, seeds=seeds, _ctx=ctx, **kw)
    got = st()
    from gaussian_process_edge_trace_amd.sequence import chain_slices
    mse = []
    for lo, hi in chain_slices(T, chains):
        want, iters = oracle_chain(frames[lo:hi], init, 16, seeds[lo:hi], kw)
        for k in range(hi - lo):
            assert st.iterations[lo + k] == iters[k], "frame %d" % (lo + k)
            assert np.array_equal(got[lo + k], want[k]), "frame %d" % (lo + k)
            mse.append(amd.gpet_utils.trace_MSE(got[lo + k], truths[lo + k]))
    assert min(st.iterations[lo + 1] for lo, hi in chain_slices(T, chains) if hi - lo > 1) >= 1  # warm frames do iterate
    print("N=%d: iterations per frame %s, MSE vs truth %s" % (N, st.iterations, np.round(mse, 1)))


def test_config5_length_64_frames_8_chains(amd, ctx):
    """BASELINE config 5 at its stated LENGTH: 64 frames at 1024x1024, Matern-5/2, 8 chains of 8 frames traced as batches
    of 8 edges (N_samples=300 to keep the CPU oracle affordable).  One chain -- the last -- is re-traced by the oracle,
    chained the same way: its first (cold) frame, its second and its last (7 warm starts deep) frame bit for bit with
    their iteration counts.  (Trace quality is whatever the algorithm gives at 300 samples on this drifting edge -- the
    oracle's and the device's traces are the same traces; it is printed, not gated.)"""
    N, T, chains = 1024, 64, 8
    frames, truths, init = make_sequence(amd, ctx, N, T)
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1,
              N_samples=300, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    seeds = [3 + t for t in range(T)]
    st = amd.SequenceTracer(frames, init, n_chains=chains, warm_every=16, seeds=seeds, _ctx=ctx, **kw)
    got = st()
    from gaussian_process_edge_trace_amd.sequence import chain_slices
    lo, hi = chain_slices(T, chains)[-1]
    assert hi - lo == 8
    want, iters = oracle_chain(frames[lo:hi], init, 16, seeds[lo:hi], kw)
    for k in (0, 1, hi - lo - 1):
        assert st.iterations[lo + k] == iters[k], "frame %d" % (lo + k)
        assert np.array_equal(got[lo + k], want[k]), "frame %d" % (lo + k)
    dice = [amd.gpet_utils.trace_dicecoef(got[t], truths[t]) for t in range(T)]
    print("64 frames / 8 chains: iterations %s, DICE min %.4f median %.4f" % (st.iterations, min(dice), float(np.median(dice))))
    assert all(np.asarray(got[t]).ndim == 2 and np.all(np.isfinite(got[t])) for t in range(T))
    assert all(st.iterations[l + k] >= 1 for l, h in chain_slices(T, chains) for k in range(1, h - l))  # warm frames do iterate


def test_sequence_single_chain_equals_per_frame_objects(amd, ctx):
    """One chain through the batch machinery (set_frame on one batch object) == a fresh GP_Edge_Tracing per frame
    with ``obs`` from the previous trace, the way a user of the reference chains frames (RBF: structured loop path,
    whose prior eigenbasis is kept across frames)."""
    from gaussian_process_edge_trace_amd.sequence import warm_start_obs
    N, T = 256, 4
    frames, truths, init = make_sequence(amd, ctx, N, T)
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 40, 'length_scale': 12}, noise_y=1, N_samples=300,
              score_thresh=1, delta_x=6, keep_ratio=0.1, pixel_thresh=4, fix_endpoints=True)
    got = amd.trace_sequence(frames, init, n_chains=1, warm_every=12, seed=5, _ctx=ctx, **kw)
    prev = None
    for t in range(T):
        obs = np.array([]) if prev is None else warm_start_obs(prev, 0, N - 1, 12, N // 6 - 3, N)
        tr = amd.GP_Edge_Tracing(init, frames[t], obs=obs, seed=5, **kw, _ctx=ctx)
        prev = tr()
        assert np.array_equal(got[t], prev), "frame %d" % t


# ---- the reference's per-method seams against the reference's own arrays -------------------------------------------
@pytest.mark.parametrize("name", ["stage_rbf64", "stage_mat128"])
def test_seams_get_best_pixels_kde_compute_new_obs(amd, ctx, golden, name):
    """get_best_curves -> kernel_density_estimate / get_best_pixels / compute_new_obs with the reference's arrays as
    ARGUMENTS (they are honoured, not replaced by device state): the reference's KDE to f32 ulps, its new observation
    set and decayed threshold exactly."""
    g = golden(name)
    L = amd._lib
    ref = g["ref_samples_head"]
    assert ref.shape[1] == int(g["ref_scalars"][2])
    tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **CTOR[name], _ctx=ctx)
    tr._batch.write(L.BUF_GRAD_KDE, g["ref_grad_kde"])  # pin the gradient KDE to the reference's
    np.testing.assert_allclose(tr.kernel_density_estimate(None, None), g["ref_grad_kde"], rtol=0, atol=0)
    curves, costs, (opt_curve, opt_cost) = tr.get_best_curves(ref)
    np.testing.assert_allclose(costs, g["ref_best_costs"], rtol=1e-9)
    np.testing.assert_allclose(opt_curve, g["ref_best_curve0"], rtol=1e-12)
    # a single curve through cost_funct leaves the scored set intact
    c0 = tr.cost_funct(g["ref_best_curve0"])
    np.testing.assert_allclose(c0, g["ref_best_costs"][0], rtol=1e-9)
    assert np.array_equal(tr._batch.read(L.BUF_BEST_IDX), g["ref_best_idxs"])
    # scramble the device's own idea of the best curves: the arguments must win
    tr._batch.write(L.BUF_BEST_IDX, np.zeros(len(costs), dtype=np.int32))
    kde = tr.kernel_density_estimate(curves, costs)
    np.testing.assert_allclose(kde, g["ref_kde_arr"], rtol=0, atol=4e-7)
    pre_yx = g["in_obs"][:, [1, 0]]
    tr._batch.write(L.BUF_BEST_IDX, np.zeros(len(costs), dtype=np.int32))
    fobs = tr.get_best_pixels(curves, costs, pre_yx)
    assert np.array_equal(fobs, g["ref_fobs"])
    assert tr.score_thresh == float(g["ref_score_thresh_out"])
    # compute_new_obs on the REFERENCE's KDE array, threshold back at its input value
    sc = tr._batch.scalars()
    sc.score_thresh = float(g["in_score_thresh"])
    tr._batch.write_scalars(sc)
    kde_ref = g["ref_kde_arr"].astype(np.float64)
    pix = np.argwhere(kde_ref > 1e-3)
    if tr.fix_endpoints:
        pix = pix[(pix[:, 1] > tr.x_st) & (pix[:, 1] < tr.x_en)]
    fobs2 = tr.compute_new_obs(pix, kde_ref, pre_yx)
    assert np.array_equal(fobs2, g["ref_fobs"])
    assert tr.score_thresh == float(g["ref_score_thresh_out"])
    with pytest.raises(ValueError):
        tr.compute_new_obs(pix[:-1], kde_ref, pre_yx)


@pytest.mark.parametrize("name", ["trace_rbf64", "trace_mat128"])
def test_seam_fit_predict_converged(amd, ctx, golden, name):
    """fit_predict_GP(obs, converged=True, seed) (gpet.py:232-248,262-266) returns the reference's optimised mean
    (pixels) and un-rescaled std for the reference's final observation set."""
    g = golden(name)
    stage = TRACES[name]
    tr = amd.GP_Edge_Tracing(g["in_init"], golden(stage)["ref_grad"], **CTOR[stage], _ctx=ctx)
    n_iter = int(g["ref_n_iter"])
    mean, std = tr.fit_predict_GP(g["ref_obs_%02d" % n_iter], converged=True, seed=tr.seed + n_iter)
    np.testing.assert_allclose(mean, g["ref_final_mean"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(mean - 1.96 * std, g["ref_ci_lower"], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("name", ["stage_rbf64", "stage_mat128"])
def test_seam_gpr_cov_and_sample_y(amd, ctx, golden, name):
    """GaussianProcessRegressor.predict(return_cov=True) and sample_y (sklearn_gpr.py:398-403, 440-473) against the
    reference's covariance and samples for the same fit (samples after aligning eigenvector signs: LAPACK's are
    implementation-defined)."""
    g = golden(name)
    N = g["ref_mean"].shape[0]
    w = np.where(np.isin(g["ref_X_train"], g["in_init"][:, 0]), 1e-7 if CTOR[name]["fix_endpoints"] else 0.5, 1.0)
    kern = {'kernel': str(g["ref_kernel_type"]), 'constant': float(g["ref_amp"]), 'length_scale': float(g["ref_sigma"][1]),
            'nu': float(g["ref_sigma"][2]),
            'white': amd.sklearn_gpr.WeightedWhiteKernel(w, N, noise_level=float(CTOR[name]["noise_y"]))}
    gp = amd.GaussianProcessRegressor(kernel=kern, alpha=1e-6, optimizer=None, normalize_y=True, _ctx=ctx)
    # y as fit_predict_GP hands it over (already divided by y_s; the fixture stores it centred)
    gp.fit(g["ref_X_train"][:, None], g["ref_y_train"] + float(g["ref_y_train_mean"]))
    xq = (np.arange(N) + int(g["in_init"][:, 0].min())).astype(float)
    mean, cov = gp.predict(xq[:, None], return_cov=True)
    np.testing.assert_allclose(mean, g["ref_mean"], rtol=1e-8)
    np.testing.assert_allclose(cov, g["ref_cov"], rtol=0, atol=1e-9 * np.abs(g["ref_cov"]).max())
    S = g["ref_samples_head"].shape[1]
    Y = gp.sample_y(xq[:, None], S, random_state=int(g["in_gp_seed"]))
    assert Y.shape == (N, S)
    # the reference's samples (pixel units there: divide by y_s) differ by the signs of the factor rows only
    F = g["ref_factor"]
    sgn = np.where(F @ (1.0 / np.arange(1, N + 1)) < 0, -1.0, 1.0)
    Z = orc.legacy_standard_normal(int(g["in_gp_seed"]), S * N).reshape(S, N)
    want = (Z * sgn[None, :]) @ F + g["ref_mean"]
    np.testing.assert_allclose(Y.T, want, rtol=0, atol=2e-6 * np.sqrt(np.abs(g["ref_cov"]).max()) + 1e-8)


# ---- T3: the device trace against the REFERENCE's own trace, as numbers ---------------------------------------------
@pytest.mark.parametrize("name", ["trace_rbf64", "trace_rbf65", "trace_mat128", "trace_mat35_96", "trace_rbf500"])
def test_t3_device_trace_vs_reference_trace(amd, ctx, golden, name):
    """The reference samples with LAPACK's singular-vector signs, the library with its documented convention
    (sum_j row[j]/(j+1) >= 0): the same posterior, different draws.  What that costs against the reference's own run
    for the same seed: iteration count within 25 %, trace within a few pixels on average, quality vs truth in the
    reference's band."""
    g = golden(name)
    stage = TRACES[name]
    tr = amd.GP_Edge_Tracing(g["in_init"], golden(stage)["ref_grad"], **CTOR[stage], _ctx=ctx)
    et = tr()
    ref = g["ref_edge_trace"]
    truth = g["in_true_edge"]
    d = np.abs(et[:, 0] - ref[:, 0])
    it_ref = int(g["ref_n_iter"])
    mse, mse_ref = amd.gpet_utils.trace_MSE(et, truth), amd.gpet_utils.trace_MSE(ref, truth)
    dice, dice_ref = amd.gpet_utils.trace_dicecoef(et, truth), amd.gpet_utils.trace_dicecoef(ref, truth)
    print("%s: |edge_trace - ref_edge_trace| max %d mean %.2f px; iterations %d vs %d; MSE vs truth %.1f (reference %.1f); "
          "DICE %.4f (reference %.4f)" % (name, d.max(), d.mean(), tr._n_iter, it_ref, mse, mse_ref, dice, dice_ref))
    assert abs(tr._n_iter - it_ref) <= max(2, it_ref // 4)
    if name == "trace_rbf500":
        # this image (seed 1) is bistable in the reference itself: its oracle lands on MSE 692 or 8443 depending on the
        # seed and on the sign convention (test_trace_quality_band); the other branch is what the library's signs select
        assert dice >= 0.85
    else:
        assert d.mean() <= 8.0  # (measured: 0.2 .. 5.5 px; different draws of the same posterior)
        assert dice >= dice_ref - 0.02


def test_t3_quality_distribution_vs_reference(amd, ctx, golden):
    """T3 on INDEPENDENT seeds (tests/golden/quality_rbf500.npz, made by tests/golden/make_quality_fixture.py): the
    README configuration on image seeds {1, 3} x 240 RNG seeds 997 apart (iteration k of seed s draws from
    RandomState(s + k + 1), so closer seeds share normal streams).
      * The device is the oracle under the library's sign convention, trace for trace: iterations, MSE and DICE of the
        480 traces are IDENTICAL to the fixture's "harmonic" rows (made with ONE BLAS thread: oracle.trace pins it) on every
        seed that the oracle itself decides -- whose harmonic row is the same with 1, 2 and 4 BLAS threads (fixture rows
        3 / 4: the oracle's own rounding noise as the perturbation); no tolerance there.  On the others (1 of 480) the device
        must give one of the oracle's answers.
      * The reference itself has no sign convention: LAPACK's singular-vector signs are implementation-defined and
        change with the BLAS thread count -- the fixture holds the oracle under LAPACK's signs with 1 and with 8 threads
        (the first equal to the unmodified reference with one thread, seed by seed), and those two disagree on most
        seeds.  Image seed 1 is bistable (a branch with MSE < 2000, one with 2000-12000); which branch a seed takes is a
        coin that every convention flips differently, so only the DISTRIBUTION is comparable: the device's good-branch
        fraction lies within 3 standard errors of each LAPACK variant (and the two LAPACK variants of each other), the
        good-branch median MSE, median DICE and iterations agree.
    The table goes to gpurun_out/r05_t3_quality.json (kept as profiles/r05_t3_quality.json)."""
    import json
    import os
    fx = golden("quality_rbf500")
    orq = fx["oracle_quality"]  # img_seed, seed, convention (0 LAPACK 1 thread, 1 harmonic 1 thread, 2 LAPACK 8 threads, 3 / 4 harmonic 2 / 4 threads), n_iter, mse, dice, relarea
    kw = CTOR["stage_rbf500"]
    seeds = sorted(set(int(v) for v in orq[orq[:, 2] == 1][:, 1]))
    assert len(seeds) >= 200 and min(np.diff(seeds)) >= 64
    report = {"config": "README: 500x500, RBF sigma_f=75 l=20, N_samples=1000, delta_x=5, pixel_thresh=5; %d RNG seeds 997 apart"
                        % len(seeds), "columns": ["n_iter", "mse", "dice"], "images": {}}

    def summary(x):
        good = x[:, 1] < 2000.0
        return {"n": int(len(x)), "n_iter_median": float(np.median(x[:, 0])), "n_iter_range": [int(x[:, 0].min()), int(x[:, 0].max())],
                "good_fraction_mse_lt_2000": float(good.mean()), "good_fraction_se": float(np.sqrt(good.mean() * (1 - good.mean()) / len(x))),
                "mse_quartiles": [float(v) for v in np.percentile(x[:, 1], [25, 50, 75])],
                "mse_median_good_branch": float(np.median(x[good, 1])),
                "dice_quartiles": [float(v) for v in np.percentile(x[:, 2], [25, 50, 75])], "dice_min": float(x[:, 2].min())}
    for img_seed in (1, 3):
        img, truth = orc.synth_sinusoid_image(500, img_seed)
        grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
        init = truth[[0, -1], :][:, [1, 0]]
        kwb = {k: v for k, v in kw.items() if k != "seed"}
        batch = amd.GP_Edge_Tracing_Batch([init] * len(seeds), np.asarray(grad, dtype=np.float32), seeds, **kwb, _ctx=ctx)
        traces = batch()
        dev = np.array([[it, amd.gpet_utils.trace_MSE(et, truth), amd.gpet_utils.trace_dicecoef(et, truth)]
                        for it, et in zip(batch.timings["iters"], traces)])
        batch._batch.close()
        rows = {}
        for conv in (0, 1, 2):
            r = orq[(orq[:, 0] == img_seed) & (orq[:, 2] == conv)]
            r = r[np.argsort(r[:, 1])]
            # (the 8-thread variant was run on the first 60 seeds only: one run at a time, five workers oversubscribe the box)
            assert [int(v) for v in r[:, 1]] == (seeds if conv < 2 else seeds[:len(r)]) and len(r) >= 60
            rows[conv] = r[:, 3:6]
        n8 = len(rows[2])
        # The device IS the oracle under the harmonic convention, seed by seed, with NO exception on every seed whose trace the
        # oracle's own arithmetic decides: the fixture holds the harmonic rows with one (the oracle's pinned default), two
        # and four BLAS threads -- the oracle perturbed by nothing but another partition of its dot products and another draw
        # of LAPACK's noise in the ~430 numerically-zero singular directions (~1e-6 px each, ~1e-7 relative on a curve's cost).
        # Where the three agree (479 of 480 traces here) the device must give exactly that row.  Where they do not, the trace
        # hangs on a near-tie below that noise (image 3, seed 179463: two costs 6e-7 apart at the N_keep cut of iteration 9)
        # and the device must give one of the oracle's answers.
        alt = {}
        for conv in (3, 4):
            rr = orq[(orq[:, 0] == img_seed) & (orq[:, 2] == conv)]
            rr = rr[np.argsort(rr[:, 1])]
            assert [int(v) for v in rr[:, 1]] == seeds
            alt[conv] = rr[:, 3:6]
        decided = np.all(rows[1] == alt[3], axis=1) & np.all(rows[1] == alt[4], axis=1)
        assert decided.sum() >= len(seeds) - 3, int(decided.sum())
        same = np.all(dev == rows[1], axis=1)
        assert np.all(same[decided]), [seeds[k] for k in np.nonzero(~same & decided)[0]]
        for k in np.nonzero(~decided)[0]:
            print("image seed %d, RNG seed %d is undecided in the oracle itself: 1 / 2 / 4 BLAS threads %s / %s / %s; device %s"
                  % (img_seed, seeds[k], rows[1][k].tolist(), alt[3][k].tolist(), alt[4][k].tolist(), dev[k].tolist()))
            assert any(np.array_equal(dev[k], v[k]) for v in (rows[1], alt[3], alt[4])), (seeds[k], dev[k])
        report.setdefault("seeds_the_oracle_itself_leaves_undecided", {})[str(img_seed)] = [seeds[k] for k in np.nonzero(~decided)[0]]
        report.setdefault("device_equals_the_oracle_on_decided_seeds", {})[str(img_seed)] = "%d of %d" % (int(same[decided].sum()), int(decided.sum()))
        sd, s0, s2 = summary(dev), summary(rows[0]), summary(rows[2])
        report["images"][str(img_seed)] = {"device = oracle, harmonic signs": sd, "oracle = reference, LAPACK signs, 1 BLAS thread": s0,
                                           "oracle, LAPACK signs, 8 BLAS threads": s2,
                                           "seeds_with_identical_quality_lapack_1_vs_8_threads": "%d of %d" % (int(np.sum(np.all(rows[0][:n8] == rows[2], axis=1))), n8),
                                           "seeds_with_identical_quality_device_vs_lapack_1_thread": int(np.sum(np.all(dev == rows[0], axis=1)))}
        print("image seed %d: device / harmonic %s" % (img_seed, sd))
        print("image seed %d: LAPACK signs, 1 thread %s" % (img_seed, s0))
        print("image seed %d: LAPACK signs, 8 threads %s" % (img_seed, s2))
        for other in (s0, s2):
            se = np.hypot(sd["good_fraction_se"], other["good_fraction_se"])
            assert abs(sd["good_fraction_mse_lt_2000"] - other["good_fraction_mse_lt_2000"]) <= 3.0 * max(se, 0.02)
            assert abs(sd["n_iter_median"] - other["n_iter_median"]) <= 1.0
            assert sd["mse_median_good_branch"] <= 1.25 * other["mse_median_good_branch"] + 25.0
            assert sd["dice_quartiles"][1] >= other["dice_quartiles"][1] - 0.01
            assert sd["dice_min"] >= other["dice_min"] - 0.05
    # the reference itself (unmodified, one BLAS thread) on its 60-seed subset = the oracle's LAPACK rows, seed by seed
    ref = fx["ref_quality_t1"]
    lut = {(int(r[0]), int(r[1])): r[3:] for r in orq if int(r[2]) == 0}
    assert len(ref) >= 100 and all(np.array_equal(lut[(int(r[0]), int(r[1]))], r[2:]) for r in ref)
    out = os.environ.get("GPET_T3_OUT", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                                                     "r05_t3_quality.json"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(report, f, indent=1)


def test_rbf_without_fix_endpoints_partial_width_takes_generic_path(amd, ctx):
    """fix_endpoints=False lets the pixel selection accept columns outside [x_st, x_en] (gpet.py:655-657 filters only
    when it is set); those are not on the grid the structured loop path indexes, so such an edge must run the generic
    path -- and still match the oracle bit for bit (round-1 advisor finding: out-of-range Q0 reads)."""
    N = 128
    img, truth = orc.synth_sinusoid_image(N, 3)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[20, 107], :][:, [1, 0]]  # partial width: x_st = 20, x_en = 107
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 20, 'length_scale': 10}, noise_y=1, N_samples=256,
              score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=3, seed=2, fix_endpoints=False)
    tr = amd.GP_Edge_Tracing(init, grad, **kw, _ctx=ctx)
    assert tr._batch.info()["structured"] == 0
    rec = []
    et_o, ci_o, info = orc.trace(init, grad, record=rec, sign_convention="harmonic", **kw)
    et, (all_samples, all_obs, curves) = tr(return_lines=True)
    assert tr._n_iter == info["n_iter"]
    for i, r in enumerate(rec):
        assert np.array_equal(all_obs[i + 1], r["obs_out"]), "iteration %d" % i
    assert np.array_equal(et, et_o)
    # the same edge with fix_endpoints=True keeps the fast path
    tr2 = amd.GP_Edge_Tracing(init, grad, **dict(kw, fix_endpoints=True), _ctx=ctx)
    assert tr2._batch.info()["structured"] == 1


def test_out_of_image_observations_are_rejected(amd, ctx):
    N = 64
    img, truth = orc.synth_sinusoid_image(N, 3)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    tr = amd.GP_Edge_Tracing(init, grad, **CTOR["stage_rbf64"], _ctx=ctx)
    for bad in ([[10, 64]], [[10, -1]], [[64, 10]], [[-1, 10]]):
        with pytest.raises(amd._lib.GpetError) as ei:
            tr._batch.set_obs(0, np.array(bad, dtype=np.int64))
        assert ei.value.code == amd._lib.ERR_BAD_ARG


DEVPTR_WORKER = r'''
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch                     # BEFORE the library: one HIP runtime per process (both have soname libamdhip64.so.7,
torch.cuda.init()                # the first one loaded serves both; torch tensors are then valid device pointers for it)
import gaussian_process_edge_trace_amd as amd
L = amd._lib
ctx = L.Context(0)
N = 128
imgs = []
for t in range(4):
    img, truth = amd.gpet_utils.construct_test_img((N, N), int(0.4 * N), 4, 0.05, 'sinusoidal', 0.3, gaps=True, seed=30 + t)
    imgs.append(amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx))
init = truth[[0, -1], :][:, [1, 0]]
kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 20, 'length_scale': 10}, noise_y=1, N_samples=256,
          score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=3, fix_endpoints=True)
dev = [torch.from_numpy(g).to("cuda:0") for g in imgs]
torch.cuda.synchronize()
host = amd.GP_Edge_Tracing_Batch([init] * 3, imgs[0], [1, 2, 3], **kw, _ctx=ctx)
onde = amd.GP_Edge_Tracing_Batch([init] * 3, None, [1, 2, 3], grad_device_ptrs=[dev[0].data_ptr()], grad_shape=(N, N), **kw, _ctx=ctx)
assert np.array_equal(host._batch.read(L.BUF_GRAD), onde._batch.read(L.BUF_GRAD))
assert np.array_equal(host._batch.read(L.BUF_GRAD_KDE), onde._batch.read(L.BUF_GRAD_KDE))
for a, b in zip(host(), onde()):
    assert np.array_equal(a, b)
host2 = amd.GP_Edge_Tracing_Batch([init] * 2, imgs[:2], [4, 5], **kw, _ctx=ctx)
onde2 = amd.GP_Edge_Tracing_Batch([init] * 2, None, [4, 5], grad_device_ptrs=[dev[0].data_ptr(), dev[1].data_ptr()], grad_shape=(N, N), **kw, _ctx=ctx)
for a, b in zip(host2(), onde2()):
    assert np.array_equal(a, b)
host2.set_frame(imgs[2:4], None, [6, 7])
onde2.set_frame(None, None, [6, 7], grad_device_ptrs=[dev[2].data_ptr(), dev[3].data_ptr()])
assert np.array_equal(host2._batch.read(L.BUF_GRAD_KDE, 1), onde2._batch.read(L.BUF_GRAD_KDE, 1))
for a, b in zip(host2(), onde2()):
    assert np.array_equal(a, b)
print("device pointers ok")
'''


def test_device_pointer_images_equal_host_images(tmp_path):
    """gpet_batch_create2 / gpet_batch_set_images with GPET_GRAD_ON_DEVICE: gradient images that already live on the GPU
    (torch CUDA tensors, the way an RCCL broadcast leaves them) are consumed in place and give what host
    arrays give -- the normalised image, the gradient KDE, every trace -- for a shared image and for one per edge, at
    construction and through set_frame.  Runs in a fresh process that imports torch FIRST, the order bench.py and the
    sharded drivers use: torch ships its own HIP runtime and a process must hold only one."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "devptr_worker.py"
    script.write_text(DEVPTR_WORKER % dict(root=root))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "device pointers ok" in r.stdout


NCCL_WORKER = r"""
import os, sys
import torch  # FIRST (INTEGRATION.md section 4)
import torch.distributed as dist
sys.path.insert(0, %(root)r)
import numpy as np
import gaussian_process_edge_trace_amd as amd
from gaussian_process_edge_trace_amd import sharding
from oracle import gpet_oracle as orc  # (synthetic image only)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
N = 128
img, truth = orc.synth_sinusoid_image(N, 2)
ctx = amd._lib.Context(0)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx).astype(np.float32)
init = truth[[0, -1], :][:, [1, 0]]
kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 20, 'length_scale': 8}, noise_y=1, N_samples=300, score_thresh=1,
          delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
inits, seeds = [init] * 6, [3, 4, 5, 6, 7, 8]
# the path trace_sharded takes with more than one rank: RCCL broadcast into a CUDA tensor (stream-synchronised by the helper),
# consumed in place through data_ptr(), traces gathered over RCCL
g = sharding.broadcast_tensor(grad, grad.shape, np.float32, dist, 0, "cuda")
assert g.is_cuda
b = amd.GP_Edge_Tracing_Batch(inits, None, seeds, grad_device_ptrs=[g.data_ptr()], grad_shape=grad.shape, _ctx=ctx, **kw)
local = b()
got = sharding.gather_traces(local, len(inits), N, dist, "cuda")
ref = amd.GP_Edge_Tracing_Batch(inits, grad, seeds, _ctx=ctx, **kw)()
assert np.array_equal(got, np.stack(ref))
dist.destroy_process_group()
print("nccl single rank ok")
"""


def test_broadcast_and_gather_over_nccl_single_rank(tmp_path):
    """The helpers trace_sharded uses with more than one rank, over a one-rank NCCL (= RCCL) group on the GPU: the gradient
    image broadcast into a CUDA tensor (broadcast_tensor synchronises torch's stream before handing the pointer to the
    library's own stream: round-2 advisor finding), consumed in place, traces gathered on the device -- equal to the
    host-array batch.  A fresh process that imports torch first."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER % dict(root=root))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "nccl single rank ok" in r.stdout
