"""Objective of the converged fits: matrix-core kernel (k_lml16, option lml_mfma=1) against the vector kernels
(k_lml / k_lml2, lml_mfma=0) and the oracle, over training-set sizes, then launch times at the bench's size.
Usage: python tools/ab_lml.py [n_train]"""
import sys
import numpy as np

sys.path.insert(0, ".")
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from tests import final_fit_inputs as ff  # noqa: E402
from oracle import gpet_oracle as orc  # noqa: E402  (checker + synthetic image)

n_time = int(sys.argv[1]) if len(sys.argv) > 1 else 98
L = amd._lib
ctx = L.Context(0)
N = 500
img, truth = orc.synth_sinusoid_image(N, 3)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
init = truth[[0, -1], :][:, [1, 0]]
rng = np.random.default_rng(0)


def training(n):
    cols = np.sort(rng.choice(np.arange(1, N - 1), size=n - 2, replace=False))
    obs = np.stack([cols, truth[cols, 0] + rng.integers(-3, 4, size=n - 2)], axis=1)
    return ff.prepare(np.asarray(init)[np.argsort(np.asarray(init)[:, 0])], obs, np.arange(N), True)


for kern, nu in (("RBF", 2.5), ("Matern", 2.5), ("Matern", 1.5), ("Matern", 0.5)):
    ko = {'kernel': kern, 'sigma_f': 75, 'length_scale': 20}
    if kern == "Matern":
        ko['nu'] = nu
    sizes = [3, 4, 5, 8, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 79, 80, 81, 95, 96, 97, 98, 99, 100, 104, 105, 108]
    kw = dict(kernel_options=ko, noise_y=1, N_samples=128, score_thresh=1, delta_x=4, keep_ratio=0.1, pixel_thresh=5,
              fix_endpoints=True)
    batch = amd.GP_Edge_Tracing_Batch([init] * len(sizes), np.asarray(grad, dtype=np.float32), list(range(len(sizes))), **kw, _ctx=ctx)
    b = batch._batch
    prs = [training(n) for n in sizes]
    for e, pr in enumerate(prs):
        b.final_set_training(e, pr["xs"], pr["yt"], pr["w"])
    reps = 4
    edge_of = np.repeat(np.arange(len(sizes), dtype=np.int32), reps)
    th = ff.BOUNDS[:, 0] + (ff.BOUNDS[:, 1] - ff.BOUNDS[:, 0]) * rng.uniform(size=(edge_of.size, 3))
    th[:, 2] = np.log(rng.uniform(1e-2, 1.0, size=edge_of.size))
    th[::reps] = np.log([5.0, 5.0, 1.0])
    L.set_option("lml_mfma", 1)
    f1, g1 = b.lml_batch(edge_of, th)
    L.set_option("lml_mfma", 0)
    f0, g0 = b.lml_batch(edge_of, th)
    L.set_option("lml_mfma", 1)
    worst = [0.0, 0.0, 0.0, 0.0]
    bad = 0
    for i, e in enumerate(edge_of):
        pr = prs[e]
        lml, g_o = orc.lml_and_grad(th[i], pr["xs"], pr["yt"], pr["w"], kern, nu)
        if not np.isfinite(lml):
            if not (np.isinf(f1[i]) and f1[i] > 0):
                bad += 1
                print("  n=%d: oracle not PD, device %r" % (sizes[e], f1[i]))
            continue
        ef = abs(f1[i] + lml) / (1 + abs(lml))
        eg = np.abs(g1[i] + g_o).max() / (1 + np.abs(g_o).max())
        ef0 = abs(f0[i] + lml) / (1 + abs(lml))
        eg0 = np.abs(g0[i] + g_o).max() / (1 + np.abs(g_o).max())
        worst = [max(worst[0], ef), max(worst[1], eg), max(worst[2], ef0), max(worst[3], eg0)]
        if not (ef < 1e-9 and eg < 1e-6):
            bad += 1
            print("  n=%d theta=%s: f %.12g vs %.12g (rel %.2e), g err %.2e   [vector kernel: %.2e %.2e]" %
                  (sizes[e], th[i], f1[i], -lml, ef, eg, ef0, eg0))
    print("%s nu=%s: %d problems, %d bad; worst rel err f %.2e g %.2e (vector kernels: %.2e %.2e)" %
          (kern, nu, edge_of.size, bad, *worst), flush=True)
    del batch

# timing at the bench's size
B = 256
kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 75, 'length_scale': 20}, noise_y=1, N_samples=128, score_thresh=1,
          delta_x=5, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
batch = amd.GP_Edge_Tracing_Batch([init] * B, np.asarray(grad, dtype=np.float32), list(range(B)), **kw, _ctx=ctx)
b = batch._batch
pr = training(n_time)
b.final_set_training_all([pr["xs"]] * B, [pr["yt"]] * B, [pr["w"]] * B)
for mode in (1, 0):
    L.set_option("lml_mfma", mode)
    for P in (1, 64, 256, 1024, 3328, 6656, 13312):
        edge_of = (np.arange(P) % B).astype(np.int32)
        th = np.tile(np.log([5.0, 5.0, 0.5]), (P, 1)) + 0.1 * rng.standard_normal((P, 3))
        f, g = b.lml_batch(edge_of, th)
        b.lml_stats(reset=True)
        for _ in range(10):
            f, g = b.lml_batch(edge_of, th)
        st = b.lml_stats()
        ms = st["kernel_ms"] / st["launches"]
        print("lml_mfma=%d n=%d P=%5d  %8.1f us/launch  %7.2f TFLOP/s (n^3 per problem)  f[0]=%.12g" %
              (mode, n_time, P, 1e3 * ms, P * n_time ** 3 / (ms * 1e-3) / 1e12, f[0]), flush=True)
