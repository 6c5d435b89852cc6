"""One launch of the matrix-core objective kernel (instrumented build prints cycles per phase). usage: python tools/prof_l16.py [n] [P]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from tests import final_fit_inputs as ff  # noqa: E402
from bench import synth_image  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 98
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = amd._lib
ctx = L.Context(0)
N = 500
img, truth = synth_image(N, 3)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
init = truth[[0, -1], :][:, [1, 0]]
rng = np.random.default_rng(0)
cols = np.sort(rng.choice(np.arange(1, N - 1), size=n - 2, replace=False))
obs = np.stack([cols, truth[cols, 0] + rng.integers(-3, 4, size=n - 2)], axis=1)
pr = ff.prepare(np.asarray(init)[np.argsort(np.asarray(init)[:, 0])], obs, np.arange(N), True)
B = 64
kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, noise_y=1, N_samples=128, score_thresh=1,
          delta_x=5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
batch = amd.GP_Edge_Tracing_Batch([init] * B, np.asarray(grad, dtype=np.float32), list(range(B)), **kw, _ctx=ctx)
b = batch._batch
b.final_set_training_all([pr["xs"]] * B, [pr["yt"]] * B, [pr["w"]] * B)
edge_of = (np.arange(P) % B).astype(np.int32)
th = np.tile(np.log([5.0, 5.0, 0.5]), (P, 1))
for _ in range(3):
    f, g = b.lml_batch(edge_of, th)
print("f[0] = %.12g" % f[0])
