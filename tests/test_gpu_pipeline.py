"""Several batch objects in flight on one GPU (pipeline.run_in_flight, what bench.py schedules): the traces are those
of the same objects run one after the other."""
import numpy as np
import pytest

from tests.test_oracle_vs_golden import CTOR

pytestmark = pytest.mark.gpu


def test_batches_in_flight_equal_sequential(golden):
    import gaussian_process_edge_trace_amd as amd
    L = amd._lib
    g = golden("stage_rbf500")
    kw = dict(CTOR["stage_rbf500"])
    kw.pop("seed")
    B = 5
    objs = [amd.GP_Edge_Tracing_Batch([g["in_init"]] * B, g["ref_grad"], seeds=list(range(10 * w + 1, 10 * w + 1 + B)), **kw,
                                      _ctx=L.Context(0)) for w in range(3)]
    seq = [[np.asarray(t) for t in o()] for o in objs]
    out = amd.run_in_flight(objs, 7)
    assert len(out) == 7
    for k, traces in enumerate(out):
        assert len(traces) == B
        for a, c in zip(traces, seq[k % 3]):
            assert np.array_equal(np.asarray(a), c), k
    for o in objs:
        o._batch.close()


def test_six_matern_batches_in_flight_equal_sequential():
    """Six batch objects whose covariances take the ANY-RANK factor (Matern: the reference's default kernel, gpet.py:25),
    all in flight at once -- six ticketed Jacobi kernels (k_oj_persist) and six blocked solves (k_chol_solve_mw) sharing
    the GPU.  Round 3's persistent Jacobi needed all its workgroups co-resident and failed the edge (GPET_ERR_STATE)
    after a 1 s barrier time-out when they were not; the ticketed form needs no residency.  The traces must equal the
    sequential runs' and the one-launch-per-round form's (option oj_persist = 0)."""
    import gaussian_process_edge_trace_amd as amd
    from oracle import gpet_oracle as orc
    L = amd._lib
    N = 512
    img, truth = orc.synth_sinusoid_image(N, 7)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)))
    init = truth[[0, -1], :][:, [1, 0]]
    warm = truth[16:-16:32][:, [1, 0]].astype(np.int64)
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1,
              N_samples=256, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    B = 3
    objs = [amd.GP_Edge_Tracing_Batch([init] * B, grad, seeds=[100 * w + 1 + 7 * e for e in range(B)], obs=[warm] * B, **kw,
                                      _ctx=L.Context(0)) for w in range(6)]
    assert objs[0]._batch.info()["structured"] == 0  # (full rank: the generic path with the any-rank factor)
    seq = [[np.asarray(t) for t in o()] for o in objs]
    out = amd.run_in_flight(objs, 12)
    for k, traces in enumerate(out):
        for a, c in zip(traces, seq[k % 6]):
            assert np.array_equal(np.asarray(a), c), k
    old = L.set_option("oj_persist", 0)
    try:
        objs[0].reset()
        rounds = [np.asarray(t) for t in objs[0]()]
    finally:
        L.set_option("oj_persist", old)
    for a, c in zip(rounds, seq[0]):
        assert np.array_equal(a, c)
    for o in objs:
        o._batch.close()
