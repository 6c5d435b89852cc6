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
