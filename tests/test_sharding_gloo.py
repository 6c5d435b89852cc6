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
