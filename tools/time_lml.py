"""Launch time of the converged fits' objective kernel at the bench's problem
size for several numbers of problems per launch.  Usage: python tools/time_lml.py [n_train]"""
import sys
import numpy as np

sys.path.insert(0, ".")
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from tests import final_fit_inputs as ff  # noqa: E402
from oracle import gpet_oracle as orc  # noqa: E402  (inputs only: synthetic image)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 98
L = amd._lib
ctx = L.Context(0)
N = 500
img, truth = orc.synth_sinusoid_image(N, 3)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
init = truth[[0, -1], :][:, [1, 0]]
B = 256
kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, noise_y=1, N_samples=1000, score_thresh=1,
          delta_x=2 if n > 100 else 5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
batch = amd.GP_Edge_Tracing_Batch([init] * B, np.asarray(grad, dtype=np.float32), list(range(B)), **kw, _ctx=ctx)
b = batch._batch
rng = np.random.default_rng(0)
cols = np.sort(rng.choice(np.arange(1, N - 1), size=n - 2, replace=False))
obs = np.stack([cols, truth[cols, 0] + rng.integers(-3, 4, size=n - 2)], axis=1)
pr = ff.prepare(np.asarray(init)[np.argsort(np.asarray(init)[:, 0])], obs, np.arange(N), True)
for e in range(B):
    b.final_set_training(e, pr["xs"], pr["yt"], pr["w"])
for two_from in (1 << 29, 1):  # one tile per thread / two tiles per thread
    L.set_option("lml_two_tiles_from", two_from)

    for P in (1, 256, 900, 3328, 6656, 13312):
        edge_of = (np.arange(P) % B).astype(np.int32)
        th = np.tile(np.log([5.0, 5.0, 0.5]), (P, 1)) + 0.1 * rng.standard_normal((P, 3))
        f, g = b.lml_batch(edge_of, th)
        b.lml_stats(reset=True)
        for _ in range(20):
            f, g = b.lml_batch(edge_of, th)
        st = b.lml_stats()
        ms = st["kernel_ms"] / st["launches"]
        print("two_tiles_from=%d n=%d P=%5d  %8.1f us/launch  %7.2f TFLOP/s (n^3 per problem)  f[0]=%.9g" %
              (two_from, n, P, 1e3 * ms, P * n ** 3 / (ms * 1e-3) / 1e12, f[0]), flush=True)
