"""world_size-2 gloo test of the multi-GPU path: contiguous edge blocks, one broadcast
of the shared gradient image, one gather; result identical to the single-process run.  The
tracer is the CPU oracle here (the GPU tracer plugs into the same callable slot)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from gaussian_process_edge_trace_amd.sharding import trace_sharded, edge_slice
from oracle import gpet_oracle as orc

KW = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 10, 'length_scale': 8}, noise_y=1, N_samples=128,
          score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=3, fix_endpoints=True)

def tracer(grad, inits, seeds):
    return [orc.trace(i, grad, seed=s, **KW)[0] for i, s in zip(inits, seeds)]

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
img, edge = orc.synth_sinusoid_image(64, 3)
init = edge[[0, -1], :][:, [1, 0]]
grad = orc.comp_grad_img(img, orc.kernel_builder((11, 5))) if rank == 0 else None   # rank 0 owns the image
n = 5
out = trace_sharded(grad, (64, 64), [init] * n, list(range(1, n + 1)), tracer, dist)
assert out.shape == (n, 64, 2)
np.save(os.path.join(%(tmp)r, "out_rank%%d.npy" %% rank), out)
dist.barrier()
dist.destroy_process_group()
'''


def test_edge_slice_partitions():
    from gaussian_process_edge_trace_amd.sharding import edge_slice
    for n in [0, 1, 5, 8, 256, 257]:
        for w in [1, 2, 3, 8]:
            cov = []
            for r in range(w):
                lo, hi = edge_slice(n, w, r)
                assert 0 <= lo <= hi <= n and hi - lo in (n // w, n // w + 1)
                cov += list(range(lo, hi))
            assert cov == list(range(n))


def test_two_rank_gloo_equals_single_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, port=port, tmp=str(tmp_path)))
    env = dict(os.environ, OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    a = np.load(tmp_path / "out_rank0.npy")
    b = np.load(tmp_path / "out_rank1.npy")
    assert np.array_equal(a, b)
    from oracle import gpet_oracle as orc
    img, edge = orc.synth_sinusoid_image(64, 3)
    init = edge[[0, -1], :][:, [1, 0]]
    grad = orc.comp_grad_img(img, orc.kernel_builder((11, 5)))
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 10, 'length_scale': 8}, noise_y=1, N_samples=128,
              score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=3, fix_endpoints=True)
    single = np.stack([orc.trace(init, grad, seed=s, **kw)[0] for s in range(1, 6)])
    assert np.array_equal(a, single)


WORKER8 = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from gaussian_process_edge_trace_amd.sharding import trace_sharded, trace_sequence_sharded, edge_slice, sequence_partition
from gaussian_process_edge_trace_amd.sequence import chain_slices, warm_start_obs
from oracle import gpet_oracle as orc

KW = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 8, 'length_scale': 8}, noise_y=1, N_samples=128,
          score_thresh=1, delta_x=6, keep_ratio=0.1, pixel_thresh=3, fix_endpoints=True)
W = %(world)d
N = 48

def tracer(grad, inits, seeds):
    return [orc.trace(i, grad, seed=s, **KW)[0] for i, s in zip(inits, seeds)]

def seq_tracer(block, first_frame, n_chains):      # chains of consecutive frames, warm-started like SequenceTracer
    out = []
    for lo, hi in chain_slices(len(block), n_chains):
        obs = np.zeros((0, 2), dtype=np.int64)
        for t in range(lo, hi):
            p = orc.resolve_params(init, np.asarray(block[t]), **KW)
            et = orc.trace(init, np.asarray(block[t]), obs=obs, seed=5 + first_frame + t, **KW)[0]
            out.append(et)
            obs = warm_start_obs(et, p["x_st"], p["x_en"], 8, p["algo_thresh"], p["M"])
    return out

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=W)
rank = dist.get_rank()
img, edge = orc.synth_sinusoid_image(N, 3)
init = edge[[0, -1], :][:, [1, 0]]
grad = orc.comp_grad_img(img, orc.kernel_builder((11, 5))) if rank == 0 else None   # rank 0 owns the image
n = 13                                                                               # 5 ranks hold 2 edges, 3 hold 1
out = trace_sharded(grad, (N, N), [init] * n, [3 + 997 * e for e in range(n)], tracer, dist)
T, C = 11, 5                                                                         # 5 chains on 8 ranks: 3 ranks idle
frames = None
if rank == 0:
    frames = np.stack([orc.comp_grad_img(orc.synth_sinusoid_image(N, 20 + t, amplitude=int(0.4 * N * (1 + 0.02 * t)))[0],
                                         orc.kernel_builder((11, 5))) for t in range(T)])
seq = trace_sequence_sharded(frames, (N, N), T, init, C, seq_tracer, dist)
lo, hi = edge_slice(n, W, rank)
f0, f1, nc = sequence_partition(T, C, W, rank)
np.savez(os.path.join(%(tmp)r, "w8_rank%%d.npz" %% rank), edges=out, seq=seq, share=np.array([hi - lo, f1 - f0, nc]))
dist.barrier()
dist.destroy_process_group()
'''


def test_eight_rank_gloo_uneven_blocks_equal_single_process(tmp_path):
    """BASELINE configs 4 and 5 are stated on 8 GPUs: the sharded tracers at world_size 8 (gloo, CPU oracle tracer) with
    blocks that do NOT divide evenly -- 13 edges (five ranks hold two, three hold one) and 11 frames in 5 chains (three
    ranks hold no chain at all): every rank ends up with the same gathered result, and it is the single-process one."""
    W = 8
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8 % dict(root=ROOT, port=port, tmp=str(tmp_path), world=W))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env) for r in range(W)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    res = [np.load(tmp_path / ("w8_rank%d.npz" % r)) for r in range(W)]
    for r in range(1, W):
        assert np.array_equal(res[r]["edges"], res[0]["edges"]) and np.array_equal(res[r]["seq"], res[0]["seq"])
    shares = np.stack([r["share"] for r in res])
    assert sorted(shares[:, 0].tolist()) == [1, 1, 1, 2, 2, 2, 2, 2] and shares[:, 0].sum() == 13
    assert shares[:, 2].sum() == 5 and (shares[:, 2] == 0).sum() == 3 and shares[:, 1].sum() == 11
    # single process
    from threadpoolctl import threadpool_limits
    from oracle import gpet_oracle as orc
    from gaussian_process_edge_trace_amd.sequence import chain_slices, warm_start_obs
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 8, 'length_scale': 8}, noise_y=1, N_samples=128,
              score_thresh=1, delta_x=6, keep_ratio=0.1, pixel_thresh=3, fix_endpoints=True)
    N = 48
    img, edge = orc.synth_sinusoid_image(N, 3)
    init = edge[[0, -1], :][:, [1, 0]]
    grad = orc.comp_grad_img(img, orc.kernel_builder((11, 5)))
    with threadpool_limits(limits=1):  # (LAPACK's eigenvector signs depend on the BLAS thread count: the workers' setting)
        single = np.stack([orc.trace(init, grad, seed=3 + 997 * e, **kw)[0] for e in range(13)])
        assert np.array_equal(res[0]["edges"], single)
        frames = [orc.comp_grad_img(orc.synth_sinusoid_image(N, 20 + t, amplitude=int(0.4 * N * (1 + 0.02 * t)))[0],
                                    orc.kernel_builder((11, 5))) for t in range(11)]
        want = []
        for lo, hi in chain_slices(11, 5):
            obs = np.zeros((0, 2), dtype=np.int64)
            for t in range(lo, hi):
                p = orc.resolve_params(init, frames[t], **kw)
                et = orc.trace(init, frames[t], obs=obs, seed=5 + t, **kw)[0]
                want.append(et)
                obs = warm_start_obs(et, p["x_st"], p["x_en"], 8, p["algo_thresh"], p["M"])
    assert np.array_equal(res[0]["seq"], np.stack(want))


def test_sequence_partition_covers_frames_in_whole_chains():
    from gaussian_process_edge_trace_amd.sharding import sequence_partition
    from gaussian_process_edge_trace_amd.sequence import chain_slices
    for T, C, W in [(64, 8, 8), (64, 8, 2), (10, 4, 3), (5, 8, 2), (7, 2, 4)]:
        chains = chain_slices(T, C)
        seen, nloc = [], 0
        for r in range(W):
            f0, f1, nc = sequence_partition(T, C, W, r)
            nloc += nc
            if nc:
                assert (f0, f1) == (chains[len(seen)][0], chains[len(seen) + nc - 1][1])
                # the local re-slicing of the block reproduces the global chain boundaries
                assert [(a + f0, b + f0) for a, b in chain_slices(f1 - f0, nc)] == chains[len(seen):len(seen) + nc]
                seen += chains[len(seen):len(seen) + nc]
        assert nloc == len(chains) and seen == chains


GPU_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
import gaussian_process_edge_trace_amd as amd
from gaussian_process_edge_trace_amd.sharding import trace_sharded, trace_sequence_sharded

KW = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 20, 'length_scale': 10}, noise_y=1, N_samples=256,
          score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=3, fix_endpoints=True)
MKW = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 20, 'length_scale': 6}, noise_y=1, N_samples=200,
           score_thresh=1, delta_x=6, keep_ratio=0.1, pixel_thresh=3, fix_endpoints=True)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
ctx = amd._lib.Context(0)          # both ranks share the box's one GPU

def tracer(grad, inits, seeds):     # the PRODUCT tracer: one batch of this rank's block of edges
    return amd.GP_Edge_Tracing_Batch(inits, np.asarray(grad), seeds, **KW, _ctx=ctx)()

N = 128
img, edge = amd.gpet_utils.construct_test_img((N, N), int(0.4 * N), 4, 0.05, 'sinusoidal', 0.3, gaps=True, seed=3)
init = edge[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx) if rank == 0 else None
n = 7
out = trace_sharded(grad, (N, N), [init] * n, list(range(1, n + 1)), tracer, dist)
np.save(os.path.join(%(tmp)r, "edges_rank%%d.npy" %% rank), out)

T = 6
frames = None
if rank == 0:
    frames = np.stack([amd.gpet_utils.comp_grad_img(
        amd.gpet_utils.construct_test_img((N, N), int(0.4 * N * (1 + 0.02 * t)), 4, 0.05, 'sinusoidal', 0.3, gaps=True, seed=20 + t)[0],
        amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx) for t in range(T)])

def seq_tracer(block, first_frame, n_chains):
    return amd.SequenceTracer(list(np.asarray(block)), init, n_chains=n_chains, warm_every=12, seed=5, _ctx=ctx, **MKW)()

seq = trace_sequence_sharded(frames, (N, N), T, init, 3, seq_tracer, dist)
np.save(os.path.join(%(tmp)r, "seq_rank%%d.npy" %% rank), seq)
dist.barrier()
dist.destroy_process_group()
'''


import pytest


@pytest.mark.gpu
def test_two_ranks_gpu_tracer_equals_single_process(tmp_path):
    """Two ranks (fresh child processes, gloo, both on device 0) run trace_sharded and trace_sequence_sharded with the
    GPU tracer: the gathered result is identical on both ranks and bit-identical to the single-process batch /
    sequence -- edges and chains are independent, so the partition must not change a pixel."""
    import gaussian_process_edge_trace_amd as amd
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "gpu_worker.py"
    script.write_text(GPU_WORKER % dict(root=ROOT, port=port, tmp=str(tmp_path)))
    env = dict(os.environ, OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    a, b = np.load(tmp_path / "edges_rank0.npy"), np.load(tmp_path / "edges_rank1.npy")
    assert np.array_equal(a, b) and a.shape == (7, 128, 2)
    sa, sb = np.load(tmp_path / "seq_rank0.npy"), np.load(tmp_path / "seq_rank1.npy")
    assert np.array_equal(sa, sb) and sa.shape == (6, 128, 2)
    # single process, one batch / one sequence tracer
    ctx = amd._lib.Context(0)
    N = 128
    img, edge = amd.gpet_utils.construct_test_img((N, N), int(0.4 * N), 4, 0.05, 'sinusoidal', 0.3, gaps=True, seed=3)
    init = edge[[0, -1], :][:, [1, 0]]
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 20, 'length_scale': 10}, noise_y=1, N_samples=256,
              score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=3, fix_endpoints=True)
    single = np.stack(amd.GP_Edge_Tracing_Batch([init] * 7, grad, list(range(1, 8)), **kw, _ctx=ctx)())
    assert np.array_equal(a, single)
    mkw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 20, 'length_scale': 6}, noise_y=1, N_samples=200,
               score_thresh=1, delta_x=6, keep_ratio=0.1, pixel_thresh=3, fix_endpoints=True)
    frames = [amd.gpet_utils.comp_grad_img(
        amd.gpet_utils.construct_test_img((N, N), int(0.4 * N * (1 + 0.02 * t)), 4, 0.05, 'sinusoidal', 0.3, gaps=True, seed=20 + t)[0],
        amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx) for t in range(6)]
    seq1 = np.stack(amd.SequenceTracer(frames, init, n_chains=3, warm_every=12, seed=5, _ctx=ctx, **mkw)())
    assert np.array_equal(sa, seq1)
