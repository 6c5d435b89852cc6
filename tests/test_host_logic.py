"""CPU tests of the host-side mirror of the reference interface: parameter clamping and its
quirks, kernel presets, the synthetic generator, metrics, the lock-step L-BFGS-B driver and the
host objective of the converged fit."""
import numpy as np
import pytest
import scipy.optimize

from gaussian_process_edge_trace_amd import gpet as G
from gaussian_process_edge_trace_amd import _final_fit as ff
from gaussian_process_edge_trace_amd._lbfgsb_lockstep import minimize_many
from oracle import gpet_oracle as orc

CASES = [
    dict(), dict(kernel_options=(0, 1, 1)), dict(kernel_options=(2, 0, 0)), dict(kernel_options=(1, 5, 4)),
    dict(kernel_options=(1, 6, 5)), dict(kernel_options={'kernel': 'Matern', 'nu': 1.5, 'sigma_f': 3, 'length_scale': 9}),
    dict(kernel_options={'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, N_samples=100, keep_ratio=0.25),
    dict(N_samples=50, keep_ratio=1.5, score_thresh=0, delta_x=3, pixel_thresh=1),
    dict(N_samples=1000, keep_ratio=0.1, delta_x=5, pixel_thresh=2, seed=1, fix_endpoints=False),
]


@pytest.mark.parametrize("kw", CASES)
def test_resolve_params_matches_oracle_and_quirks(kw):
    grad = np.zeros((120, 200), np.float32)
    init = np.array([[190, 60], [10, 50]])  # deliberately unsorted (Q4)
    p = G.resolve_params(init, grad.shape, **kw)
    o = orc.resolve_params(init, grad, **kw)
    for a, b in [("x_st", "x_st"), ("x_en", "x_en"), ("N_samples", "N_samples"), ("N_keep", "N_keep"),
                 ("N_subints", "N_subints"), ("algo_thresh", "algo_thresh"), ("delta_x", "delta_x"),
                 ("pixel_thresh", "pixel_thresh"), ("score_thresh", "score_thresh"), ("keep_ratio", "keep_ratio"),
                 ("sigma_f", "sigma_f"), ("sigma_l", "length_scale"), ("kernel_type", "kernel_type"),
                 ("kernel_nu", "nu"), ("edge_length", "edge_length")]:
        assert p[a] == o[b], (a, p[a], o[b])
    assert np.array_equal(p["init"], o["init"]) and np.array_equal(p["x_grid"], o["x_grid"])


def test_positional_readme_call_binds_like_the_reference():
    """README.md:75-76 predates pixel_thresh: seed=1 lands in pixel_thresh, return_std in seed (Q6)."""
    import inspect
    sig = list(inspect.signature(G.GP_Edge_Tracing.__init__).parameters)
    assert sig[:14] == ["self", "init", "grad_img", "kernel_options", "noise_y", "obs", "N_samples", "score_thresh",
                        "delta_x", "keep_ratio", "pixel_thresh", "seed", "return_std", "fix_endpoints"]
    call = list(inspect.signature(G.GP_Edge_Tracing.__call__).parameters)
    assert call[:6] == ["self", "print_final_diagnostics", "show_init_post", "show_post_iter", "verbose", "return_lines"]


def test_abi_params_and_factor_cap():
    p = G.resolve_params(np.array([[0, 250], [499, 250]]), (500, 500),
                         {'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, 1, np.array([]), 1000, 1, 5, 0.1, 5, 1)
    q = G.to_abi_params(p)
    assert (q.n_samples, q.n_keep, q.delta_x, q.pixel_thresh, q.x_st, q.x_en, q.n_init) == (1000, 100, 5, 5, 0, 499, 2)
    assert q.jitter == 1e-6 and q.factor_cap == 0  # 2.6*500/20+12 = 77 <= 96 -> LDS Jacobi path
    p2 = dict(p, kernel_type="Matern")
    assert G.auto_factor_cap(p2) == 500
    p3 = dict(p, sigma_l=5.0)
    assert 96 < G.auto_factor_cap(p3) <= 500


def test_kernel_builder_and_synthetic_generator_match_oracle():
    from gaussian_process_edge_trace_amd import gpet_utils as U
    for sz in [(11, 5), (5, 3), (7, 7), (3, 4)]:
        for kw in [{}, {'b2d': True}, {'unit': True}, {'vertical_edges': True}, {'normalize': True}]:
            assert np.array_equal(U.kernel_builder(sz, **kw), orc.kernel_builder(sz, **kw))
    img, edge = U.construct_test_img((500, 500), 200, 4, 0.05, 'sinusoidal', 0.3, gaps=True, seed=1)
    img_o, edge_o = orc.synth_sinusoid_image(500, 1)
    assert np.array_equal(img, img_o) and np.array_equal(edge, edge_o)
    assert U.trace_MSE(edge, edge) == 0 and U.trace_dicecoef(edge, edge) == 1.0 and U.trace_relarea(edge, edge) == 0
    shifted = edge.copy()
    shifted[:, 0] += 3
    assert U.trace_MSE(shifted, edge) == 9.0 and 0.9 < U.trace_dicecoef(shifted, edge) < 1.0


def test_lockstep_lbfgsb_equals_scipy_minimize():
    rng = np.random.default_rng(0)
    bounds = np.array([[-2.0, 2.0], [-1.0, 3.0], [0.5, 2.5]])

    def fg(x):
        f = (1 - x[0]) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2 + (x[2] - 1.2) ** 4
        g = np.array([-2 * (1 - x[0]) - 400 * x[0] * (x[1] - x[0] ** 2), 200 * (x[1] - x[0] ** 2), 4 * (x[2] - 1.2) ** 3])
        return f, g

    x0s = [rng.uniform(bounds[:, 0], bounds[:, 1]) for _ in range(9)] + [np.array([5.0, -7.0, 0.0])]
    ref = [scipy.optimize.minimize(fg, x0, method="L-BFGS-B", jac=True, bounds=bounds) for x0 in x0s]

    def eval_batch(idx, X):
        out = [fg(x) for x in X]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out])

    X, F, rounds = minimize_many(eval_batch, x0s, bounds)
    for i, r in enumerate(ref):
        assert np.array_equal(X[i], r.x) and F[i] == r.fun
    assert rounds == max(r.nfev for r in ref)


def test_host_final_fit_matches_oracle(golden):
    g = golden("trace_rbf64")
    init = g["in_init"][np.argsort(g["in_init"][:, 0])]
    obs = g["ref_obs_%02d" % int(g["ref_n_iter"])]
    xg = np.arange(init[0, 0], init[-1, 0] + 1)
    mean, std, theta = ff.converged_fit_predict(init, obs, xg, "RBF", 2.5, 1, True, 1 + int(g["ref_n_iter"]))
    p = dict(fix_endpoints=True, x_grid=xg, kernel_type="RBF", nu=2.5, noise_y=1)
    mo, so, info = orc.converged_fit_predict(init, obs, p, 1 + int(g["ref_n_iter"]))
    np.testing.assert_allclose(theta, info["theta"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(mean, mo, rtol=1e-9)
    np.testing.assert_allclose(theta, g["ref_final_theta"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mean, g["ref_final_mean"], rtol=1e-6, atol=1e-6)


def _quartic_problems(seed, P):
    r = np.random.default_rng(seed)
    A, c, x0 = r.uniform(0.5, 5, (P, 3)), r.uniform(-2, 2, (P, 3)), r.uniform(-3, 3, (P, 3))

    def ev(idx, X):
        d = X - c[idx]
        return np.sum(A[idx] * d ** 4 + d * d, axis=1), 4 * A[idx] * d ** 3 + 2 * d
    return list(x0), ev


def test_lockstep_farm_two_concurrent_jobs_equal_the_in_process_driver():
    """Worker processes shared by two job slots (as the pipelined bench uses them): each job follows exactly the
    iterates of the in-process lock-step driver (= scipy.optimize.minimize, previous test), also when a slot is reused."""
    import threading
    from gaussian_process_edge_trace_amd._lbfgsb_lockstep import LockstepFarm, minimize_many
    bounds = np.array([[-3.0, 3.0]] * 3)
    farm = LockstepFarm(3, slots=2)
    try:
        res = {}

        def run(slot, seed, P):
            x0, ev = _quartic_problems(seed, P)
            res[slot] = farm.slot(slot).minimize(ev, x0, bounds)
        ts = [threading.Thread(target=run, args=(0, 1, 40)), threading.Thread(target=run, args=(1, 2, 55))]
        [t.start() for t in ts]
        [t.join(timeout=60) for t in ts]
        assert not any(t.is_alive() for t in ts)
        run(0, 3, 20)  # slot reuse
        for slot, (seed, P) in {1: (2, 55), 0: (3, 20)}.items():
            x0, ev = _quartic_problems(seed, P)
            Xr, Fr, _ = minimize_many(ev, x0, bounds)
            X, F, rounds = res[slot]
            assert rounds > 3 and np.array_equal(X, Xr) and np.array_equal(F, Fr)
    finally:
        farm.close()


def test_committed_bench_line_keeps_the_driver_contract():
    """profiles/r01_bench_n1.json is a bench.py output committed this round: it must carry every field of the
    driver's contract (metric/value/unit/..., roofline{...}, cpu_baseline{...}) with sane types."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r01_bench_n1.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "edge-traces/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    assert d["value"] > 50 * c["value"]  # north_star target: >= 50x the CPU path on one GPU


def test_prepare_many_is_bit_identical_to_prepare():
    """The grouped (stacked, axis=1) standardisation of many training sets equals the per-edge one bit for bit:
    same argsort permutation, means, stds and standardised values, including a set that fills the whole x-grid
    (weights zeroed) and sets with repeated x."""
    from gaussian_process_edge_trace_amd import _final_fit as ff
    rng = np.random.default_rng(5)
    inits, obs, grids, fes = [], [], [], []
    for e in range(40):
        n_obs = [30, 31, 30, 97, 30][e % 5]
        xg = np.arange(0, 500) if e % 7 else np.arange(0, n_obs + 2)
        x = rng.choice(np.arange(1, len(xg) - 1), size=n_obs, replace=(e % 3 == 0)) if len(xg) > n_obs + 2 else np.arange(1, n_obs + 1)
        inits.append(np.array([[0, rng.integers(0, 400)], [len(xg) - 1, rng.integers(0, 400)]]))
        obs.append(np.stack([x, rng.integers(0, 400, size=n_obs)], axis=1).astype(np.int64))
        grids.append(xg)
        fes.append(bool(e % 2))
    many = ff.prepare_many(inits, obs, grids, fes)
    for e in range(40):
        one = ff.prepare(inits[e], obs[e], grids[e], fes[e])
        for k in ("xs", "yt", "w"):
            assert np.array_equal(many[e][k], one[k]), (e, k)
        for k in ("y_m", "y_s", "X_m", "X_s", "m2", "s2"):
            assert many[e][k] == one[k], (e, k)


def test_start_points_many_equals_randomstate_per_edge():
    """Vectorised legacy seeding (init_genrand for all seeds at once) + one reused RandomState: the restart points
    equal ``start_points`` (= ``RandomState(seed).uniform``) bit for bit, including the extreme seeds."""
    from gaussian_process_edge_trace_amd import _final_fit as ff
    seeds = [0, 1, 2, 42, 12345, 2**31 - 1, 2**31, 2**32 - 1] + list(range(100, 140))
    noise = [1.0, 0.5] * (len(seeds) // 2)
    many = ff.start_points_many(noise, seeds)
    k = 0
    for nz, sd in zip(noise, seeds):
        for th in ff.start_points(nz, sd):
            assert np.array_equal(many[k], th), (sd, k)
            k += 1
    assert k == len(many)
