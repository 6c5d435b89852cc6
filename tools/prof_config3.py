"""Config 3's fit + predict + covariance (n = 1500, Lg = 2048) a few times, for rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_edge_trace_amd as amd
from bench import synth_image
ctx = amd._lib.Context(0)
N = 2048
img, truth = synth_image(N, 0)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
init = truth[[0, -1], :][:, [1, 0]]
rng = np.random.default_rng(0)
cols = np.sort(rng.choice(np.arange(1, N - 1), size=1498, replace=False))
obs = np.stack([cols, np.clip(truth[cols, 0] + rng.integers(-2, 3, size=cols.size), 0, N - 1)], axis=1).astype(np.int64)
kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 300, 'length_scale': 80}, noise_y=1, N_samples=4000,
          score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, seed=1, fix_endpoints=True)
tr = amd.GP_Edge_Tracing(init, grad, obs=obs, **kw, _ctx=ctx)
b = tr._batch
b.set_obs(0, obs)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
import time
for _ in range(reps):
    ctx.sync(); t0 = time.time(); b.fit_predict(True); ctx.sync(); print("fit+predict+cov %.2f ms" % (1e3 * (time.time() - t0)))
b.factor(); b.normals([7]); b.sample(); b.score()
