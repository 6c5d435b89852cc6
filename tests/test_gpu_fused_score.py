"""The opt-in fused form of a6 + a7 in the device loop (gpet_set_option "fused_score": k_sample_score +
k_sample_keep_rows -- the samples of an iteration are scored out of the matrix-core accumulators and only the N_keep best
rows are stored) against the separate sample GEMM and scorer (the default): same sample bits, same costs, same best
indices, same kept rows, same traces (sklearn_gpr.py:440-473, gpet.py:391-451)."""
import numpy as np
import pytest

from tests.test_oracle_vs_golden import CTOR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import gaussian_process_edge_trace_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def ctx(amd):
    return amd._lib.Context(0)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("name,dtype", [("stage_rbf500", "f64"), ("stage_rbf500", "f32"), ("stage_rbf64", "f64")])
def test_fused_stage_equals_gemm_plus_scorer(amd, ctx, golden, name, dtype, variant):
    """One iteration's state (structured factor, numpy-stream normals): costs from the accumulators == costs of the
    stored samples, bit for bit; the kept rows are the GEMM's rows; no other row is written."""
    L = amd._lib
    g = golden(name)
    tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **CTOR[name], sample_dtype=dtype, _ctx=ctx)
    b = tr._batch
    info = b.info()
    assert info["structured"] == 1 and info["Lg"] % 2 == 0
    b.set_obs(0, g["in_obs"])
    for stage in (120, 121, 122, 123):
        b.profile_stage(stage, 1)
    b.normals([int(g["in_gp_seed"])])
    # separate kernels
    b.profile_stage(130, 1)
    b.profile_stage(140, 1)
    b.profile_stage(141, 1)
    Y_u = b.read(L.BUF_SAMPLES)
    costs_u, idx_u, best_u = b.read(L.BUF_COSTS), b.read(L.BUF_BEST_IDX), b.read(L.BUF_BEST_COSTS)
    assert np.isfinite(costs_u).all() and len(set(idx_u.tolist())) == idx_u.shape[0]
    # fused: poison what it must overwrite
    marker = -12345.0
    b.write(L.BUF_SAMPLES, np.full_like(Y_u, marker))
    b.write(L.BUF_COSTS, np.full_like(costs_u, np.nan))
    b.profile_stage(132, 1)  # (best_idx is still the separate scorer's)
    Y_f = b.read(L.BUF_SAMPLES)
    assert np.array_equal(Y_f[idx_u], Y_u[idx_u]), np.abs(Y_f[idx_u] - Y_u[idx_u]).max()
    old = L.set_option("fused_score", variant)  # (1: column-tile-stationary k_sample_score, 2: curve-stationary k_sample_score2)
    try:
        b.profile_stage(131, 1)
    finally:
        L.set_option("fused_score", old)
    costs_f = b.read(L.BUF_COSTS)
    assert np.array_equal(costs_f, costs_u), np.abs(costs_f / costs_u - 1).max()
    b.profile_stage(141, 1)
    assert np.array_equal(b.read(L.BUF_BEST_IDX), idx_u) and np.array_equal(b.read(L.BUF_BEST_COSTS), best_u)
    rest = np.setdiff1d(np.arange(Y_u.shape[0]), idx_u)
    assert (Y_f[rest] == marker).all()


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("name", ["stage_rbf64", "stage_mat15_96", "stage_rbf500"])
def test_fused_loop_traces_equal_unfused(amd, ctx, golden, name, variant):
    """Whole traces with the fused kernel (default) and with the separate GEMM + scorer: identical observation sets per
    iteration, iteration counts, final costs and edge traces; return_lines hands back every sample in both modes."""
    L = amd._lib
    g = golden(name)
    kw = dict(CTOR[name])
    out = {}
    for fused in (1, 0):
        old = L.set_option("fused_score", variant if fused else 0)
        try:
            tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **kw, _ctx=ctx)
            et = tr()
            out[fused] = (et, tr._n_iter, tr._batch.read(L.BUF_COSTS), tr._batch.read(L.BUF_BEST_IDX),
                          tr._batch.read(L.BUF_OBS))
        finally:
            L.set_option("fused_score", old)
    for a, c in zip(out[1], out[0]):
        assert np.array_equal(np.asarray(a), np.asarray(c))
    # return_lines: the whole sample matrix of every iteration (the loop then runs the separate kernels)
    old = L.set_option("fused_score", variant)
    try:
        tr = amd.GP_Edge_Tracing(g["in_init"], g["ref_grad"], **kw, _ctx=ctx)
        et, (all_samples, all_obs, curves) = tr(return_lines=True)
    finally:
        L.set_option("fused_score", old)
    assert np.array_equal(et, out[1][0]) and len(all_samples) >= out[1][1] and all_samples[0].shape[1] == kw["N_samples"]
    for Y in all_samples:
        assert np.isfinite(Y).all() and np.abs(Y).max() > 0 and (np.abs(Y).max(axis=0) > 0).all()


@pytest.mark.parametrize("variant", [1, 2])
def test_fused_loop_batch_of_edges(amd, ctx, golden, variant):
    """A batch of edges with their own seeds (config 4's form): fused == separate for every edge."""
    L = amd._lib
    g = golden("stage_rbf500")
    kw = dict(CTOR["stage_rbf500"])
    kw.pop("seed")
    B = 6
    res = {}
    for fused in (1, 0):
        old = L.set_option("fused_score", variant if fused else 0)
        try:
            bt = amd.GP_Edge_Tracing_Batch([g["in_init"]] * B, g["ref_grad"], seeds=list(range(3, 3 + B)), **kw, _ctx=ctx)
            traces = bt()
            res[fused] = ([np.asarray(t) for t in traces], list(bt.timings["iters"]))
            bt._batch.close()
        finally:
            L.set_option("fused_score", old)
    assert res[1][1] == res[0][1]
    for a, c in zip(res[1][0], res[0][0]):
        assert np.array_equal(a, c)
