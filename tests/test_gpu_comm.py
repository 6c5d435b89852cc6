"""The C ABI's collectives (include/gpet_hip.h: gpet_comm_*, gpet_bcast_grad, gpet_gather_traces -- SURVEY 8b / 8e) through
ctypes, no torch in the process: a world of one on the box's GPU, and two ranks over RCCL where the host has two GPUs."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 20, 'length_scale': 10}, noise_y=1, N_samples=256,
          score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=3, fix_endpoints=True)


def _problem(amd, ctx, N=128):
    img, edge = amd.gpet_utils.construct_test_img((N, N), int(0.4 * N), 4, 0.05, 'sinusoidal', 0.3, gaps=True, seed=3)
    init = edge[[0, -1], :][:, [1, 0]]
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    return grad, init


@pytest.mark.gpu
@pytest.mark.parametrize("through_rccl", [0, 1])
def test_world_of_one_through_the_c_abi(through_rccl):
    """through_rccl = 1: the communicator and both collectives are real RCCL calls (ncclCommInitRank with one rank,
    ncclBroadcast, ncclAllGather) -- the library's run-time binding of librccl is exercised on any box."""
    import gaussian_process_edge_trace_amd as amd
    from gaussian_process_edge_trace_amd.sharding import trace_sharded_cabi
    L = amd._lib
    ctx = L.Context(0)
    grad, init = _problem(amd, ctx)
    old = L.set_option("comm_force_rccl", through_rccl)
    try:
        comm = L.Comm(ctx, None, 1, 0)
    finally:
        L.set_option("comm_force_rccl", old)
    assert (comm.rank, comm.world) == (0, 1) and comm.block(7) == (0, 7)
    ptr = comm.bcast_grad(grad, grad.shape)
    assert np.array_equal(comm.download(ptr, grad.shape), np.asarray(grad, dtype=np.float32))
    seeds = list(range(1, 8))
    got = trace_sharded_cabi(grad, grad.shape, [init] * 7, seeds, comm, **KW)
    want = np.stack(amd.GP_Edge_Tracing_Batch([init] * 7, grad, seeds, **KW, _ctx=ctx)())
    assert got.dtype == np.int64 and np.array_equal(got, want)
    assert np.array_equal(comm.allgather_i64(np.arange(5), [5]), np.arange(5))
    with pytest.raises(L.GpetError):
        L.Comm(ctx, None, 2, 0)  # (a world of two needs the unique id)
    comm.close()


WORKER = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
import gaussian_process_edge_trace_amd as amd
from gaussian_process_edge_trace_amd.sharding import trace_sharded_cabi
rank, world, tmp = int(sys.argv[1]), 2, %(tmp)r
L = amd._lib
ctx = L.Context(rank)                      # one process per GPU
idf = os.path.join(tmp, "rccl_id.bin")
if rank == 0:
    uid = L.comm_unique_id()
    open(idf + ".tmp", "wb").write(uid)
    os.replace(idf + ".tmp", idf)          # (any transport will do: here a file)
else:
    t0 = time.time()
    while not os.path.exists(idf):
        assert time.time() - t0 < 120
        time.sleep(0.05)
    uid = open(idf, "rb").read()
comm = L.Comm(ctx, uid, world, rank)
N = 128
KW = %(kw)r
grad = init = None
img, edge = amd.gpet_utils.construct_test_img((N, N), int(0.4 * N), 4, 0.05, 'sinusoidal', 0.3, gaps=True, seed=3)
init = edge[[0, -1], :][:, [1, 0]]
if rank == 0:
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
out = trace_sharded_cabi(grad, (N, N), [init] * 7, list(range(1, 8)), comm, **KW)
np.save(os.path.join(tmp, "cabi_rank%%d.npy" %% rank), out)
cnt = [3, 5]
ag = comm.allgather_i64(np.arange(cnt[rank]) + 100 * rank, cnt)
assert np.array_equal(ag, np.concatenate([np.arange(3), np.arange(5) + 100])), ag
comm.close()
'''


@pytest.mark.gpu
def test_two_ranks_over_rccl_through_the_c_abi(tmp_path):
    """Two processes, one GPU each, RCCL bound by the library itself: broadcast of the gradient image into device memory,
    blocks of edges, gather of the traces -- equal on both ranks and to the single-process batch."""
    import gaussian_process_edge_trace_amd as amd
    L = amd._lib
    try:
        L.Context(1).close()
    except L.GpetError:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    script = tmp_path / "cabi_worker.py"
    script.write_text(WORKER % dict(root=ROOT, tmp=str(tmp_path), kw=KW))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    a, b = np.load(tmp_path / "cabi_rank0.npy"), np.load(tmp_path / "cabi_rank1.npy")
    assert np.array_equal(a, b) and a.shape == (7, 128, 2)
    ctx = L.Context(0)
    grad, init = _problem(amd, ctx)
    assert np.array_equal(a, np.stack(amd.GP_Edge_Tracing_Batch([init] * 7, grad, list(range(1, 8)), **KW, _ctx=ctx)()))
