"""Ad-hoc: device time of one GP iteration + scoring at BASELINE config 3 (2048^2, n = 1500, S = 4000)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import gaussian_process_edge_trace_amd as amd
    from oracle import gpet_oracle as orc  # (image generator only)
    ctx = amd._lib.Context(0)
    N = 2048
    img, truth = orc.synth_sinusoid_image(N, 0)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    rng = np.random.default_rng(0)
    cols = np.sort(rng.choice(np.arange(1, N - 1), size=1498, replace=False))
    obs = np.stack([cols, truth[cols, 0] + rng.integers(-2, 3, size=cols.size)], axis=1).astype(np.int64)
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 300, 'length_scale': 80}, noise_y=1, N_samples=4000,
              score_thresh=1, delta_x=5, keep_ratio=0.1, pixel_thresh=5, seed=1, fix_endpoints=True)
    t0 = time.time()
    tr = amd.GP_Edge_Tracing(init, grad, obs=obs, **kw, _ctx=ctx)
    b = tr._batch
    print("ctor %.3f s; info %s" % (time.time() - t0, b.info()))
    b.set_obs(0, obs)
    for name, fn in [("fit+predict+cov", lambda: b.fit_predict(want_cov=True)), ("factor", b.factor),
                     ("normals", lambda: b.normals([7])), ("sample", b.sample), ("score", b.score)]:
        for rep in range(2):
            ctx.sync(); ts = time.time(); fn(); ctx.sync(); dt = time.time() - ts
        print("%-16s %.1f ms" % (name, 1e3 * dt), flush=True)




def structured_large_n():
    """The loop's structured path at ~1000 on-grid training points (delta_x = 2 keeps the edge unfinished)."""
    import gaussian_process_edge_trace_amd as amd
    from oracle import gpet_oracle as orc
    ctx = amd._lib.Context(0)
    N = 2048
    img, truth = orc.synth_sinusoid_image(N, 0)
    grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    rng = np.random.default_rng(1)
    cols = np.sort(rng.choice(np.arange(1, N - 1), size=1000, replace=False))
    obs = np.stack([cols, truth[cols, 0] + rng.integers(-2, 3, size=cols.size)], axis=1).astype(np.int64)
    kw = dict(kernel_options={'kernel': 'RBF', 'sigma_f': 300, 'length_scale': 80}, noise_y=1, N_samples=4000,
              score_thresh=1, delta_x=2, keep_ratio=0.1, pixel_thresh=5, seed=1, fix_endpoints=True)
    tr = amd.GP_Edge_Tracing(init, grad, **kw, _ctx=ctx)
    b = tr._batch
    b.set_obs(0, obs)
    print("structured loop path, n = 1002, Lg = 2048, S = 4000:", b.info())
    tot = 0.0
    for sid, name in [(120, "fit (blocked Cholesky + solves)"), (121, "B, U = L^-1 B, H, mean"), (122, "Jacobi"),
                      (123, "factor rows"), (130, "sample GEMM"), (140, "scoring"), (141, "top-k")]:
        ms = b.profile_stage(sid, 5)
        tot += ms
        print("  %-34s %.3f ms" % (name, ms), flush=True)
    print("  sum %.2f ms per GP iteration + scoring" % tot)


if __name__ == "__main__":
    main()
    structured_large_n()
