"""CPU tests of the host-side mirror of the reference interface: parameter clamping and its
quirks, kernel presets, the synthetic generator, metrics, the lock-step L-BFGS-B driver and the
host objective of the converged fit."""
import numpy as np
import pytest

from gaussian_process_edge_trace_amd import gpet as G
from tests import final_fit_inputs as ff
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


def test_multi_sinusoidal_generators_follow_the_reference_recipe():
    """The two two-edge image types (gpet_utils.py:203-220): per column the first edge at rint(A sin) + M//2, the second
    A//2 (resp. A//6) rows below it, the band under the second edge at 1 - intensity; the truth holds both edges, one
    after the other.  Checked against the recipe written out column by column (no noise)."""
    from gaussian_process_edge_trace_amd import gpet_utils as U
    M = N = 96
    for ltype, div in (("multi-sinusoidal", 2), ("close multi-sinusoidal", 6)):
        img, edge = U.construct_test_img((M, N), 40, 4, 0.0, ltype, 0.3, gaps=True, seed=1)
        A = 40 // 2
        x = np.linspace(-np.pi, np.pi, N)
        ref = np.zeros((M, N))
        y0 = []
        for j in range(N):
            w = int(np.rint(A * np.sin(N * 4 * x[j])) + M // 2)
            y0.append(w)
            ref[w:M, j] = 0.3
            ref[w + A // div:M, j] = 1 - 0.3
        for a, b in ((20, 30), (N // 2, N // 2 + 10), (N - 100, N - 90), (N // 4, N // 4 + 20)):
            ref[:, a:b] = 0  # (N - 100 < 0 here: the empty slice of the reference)
        assert np.array_equal(img, ref)
        assert edge.shape == (2 * N, 2)
        assert np.array_equal(edge[:N, 0], y0) and np.array_equal(edge[N:, 0], np.asarray(y0) + A // div)
        assert np.array_equal(edge[:N, 1], np.arange(N)) and np.array_equal(edge[N:, 1], np.arange(N))
    with pytest.raises(ValueError):
        U.construct_test_img((M, N), 40, 4, 0.0, "zigzag", 0.3)


def test_host_start_points_are_the_reference_restarts():
    """``start_points`` = theta of the kernel (gpet.py:244-245) + RandomState(seed).uniform over the log-bounds
    (sklearn_gpr.py:283-288) -- the oracle draws the same numbers from its own MT19937."""
    for seed in (0, 1, 42, 2**32 - 1):
        th = np.asarray(ff.start_points(0.5, seed))
        u = orc.legacy_uniform(seed, 36).reshape(12, 3)
        assert np.array_equal(th[1:], ff.BOUNDS[:, 0] + (ff.BOUNDS[:, 1] - ff.BOUNDS[:, 0]) * u)
        assert np.array_equal(th[0], np.log([5.0, 5.0, 0.5]))


def test_committed_bench_line_keeps_the_driver_contract():
    """profiles/rNN_bench_n1.json (the latest round's) is a bench.py output committed with the code: it must carry every
    field of the driver's contract (metric/value/unit/..., roofline{...}, cpu_baseline{...}) with sane types."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import glob
    d = json.load(open(sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_bench_n1.json")))[-1]))
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


def test_run_in_flight_schedules_every_step_once_and_in_order():
    """pipeline.run_in_flight: step k on object k mod W, every step exactly once, results in step order, an object only
    ever inside one call at a time, a worker's exception re-raised (host logic only: fake tracers)."""
    import threading
    import time as _t
    from gaussian_process_edge_trace_amd.pipeline import run_in_flight, step_owner

    assert step_owner(10, 3) == [[0, 3, 6, 9], [1, 4, 7], [2, 5, 8]]
    assert step_owner(2, 6) == [[0], [1]]

    class Fake:
        def __init__(self, name):
            self.name, self.busy, self.calls, self.lock = name, 0, [], threading.Lock()

        def reset(self):
            self.calls.append("reset")

        def __call__(self):
            with self.lock:
                self.busy += 1
                assert self.busy == 1
            _t.sleep(0.002)
            with self.lock:
                self.busy -= 1
            self.calls.append("trace")
            return self.name

    objs = [Fake("a"), Fake("b"), Fake("c")]
    out = run_in_flight(objs, 8)
    assert out == ["a", "b", "c", "a", "b", "c", "a", "b"]
    assert [o.calls.count("trace") for o in objs] == [3, 3, 2] and all(o.calls[0] == "reset" for o in objs)
    seen = []
    out = run_in_flight(objs[:2], 3, prepare=lambda tr, k: seen.append((tr.name, k)))
    assert out == ["a", "b", "a"] and sorted(seen) == [("a", 0), ("a", 2), ("b", 1)]

    class Bad(Fake):
        def __call__(self):
            raise RuntimeError("boom")

    import pytest
    with pytest.raises(RuntimeError):
        run_in_flight([Fake("a"), Bad("b")], 4)
