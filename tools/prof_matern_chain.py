"""One chain of Matern-5/2 frames (config 5's per-GPU share) for rocprofv3 --kernel-trace --stats.
usage: python tools/prof_matern_chain.py [frames] [chains]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import gaussian_process_edge_trace_amd as pkg  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = 1024
ctx = pkg._lib.Context(0)
frames = []
for t in range(T * C):
    img, truth = pkg.gpet_utils.construct_test_img((N, N), int(0.4 * N * (1.0 + 0.01 * (t % T))), 4, 0.05, 'sinusoidal', 0.3, gaps=True, seed=100 + t)
    frames.append(pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx))
init = truth[[0, -1], :][:, [1, 0]]
kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 154, 'length_scale': 41}, noise_y=1, N_samples=1000,
          score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
st = pkg.SequenceTracer(frames, init, n_chains=C, warm_every=16, seed=3, _ctx=ctx, **kw)
t0 = time.time()
st()
print("%d frames in %d chain(s): %.3f s, iterations %s" % (T * C, C, time.time() - t0, st.iterations))
