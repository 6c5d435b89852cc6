"""Timing of the full-rank (Matern) path at BASELINE config 5's frame shape: factor ms for one edge and for a batch,
whole frame on the GPU and in the CPU oracle.  usage: python tools/time_matern.py [N] [B] [--oracle]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    N = int(args[0]) if len(args) > 0 else 1024
    B = int(args[1]) if len(args) > 1 else 8
    import gaussian_process_edge_trace_amd as pkg
    from oracle import gpet_oracle as orc
    ctx = pkg._lib.Context(0)
    img, truth = orc.synth_sinusoid_image(N, 5)
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    warm = truth[16:-16:16][:, [1, 0]].astype(np.int64)
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1,
              N_samples=1000, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    tr = pkg.GP_Edge_Tracing(init, grad, obs=warm, seed=3, **kw, _ctx=ctx)
    b = tr._batch
    b.set_obs(0, warm)
    t = time.time(); b.iterate([3], 1); dt = time.time() - t
    s = b.scalars()
    print("N=%d first iteration %.3f s rank=%d sweeps=%d n=%d" % (N, dt, s.rank, int(s.lml), s.n), flush=True)
    for i, name in enumerate(["fit_predict_cov", "factor", "normals", "gemm", "score", "kde"]):
        print("  single edge %-16s %.2f ms" % (name, b.profile_stage(i, 3)), flush=True)
    t = time.time(); tr(); print("single-edge frame (warm start): %.3f s, %d iterations" % (time.time() - t, tr._n_iter), flush=True)
    cold = pkg.GP_Edge_Tracing(init, grad, seed=3, **kw, _ctx=ctx)
    t = time.time(); cold(); print("single-edge frame (cold start): %.3f s, %d iterations" % (time.time() - t, cold._n_iter), flush=True)
    del tr, cold
    if B > 1:
        bt = pkg.GP_Edge_Tracing_Batch([init] * B, grad, [3 + e for e in range(B)], **kw, _ctx=ctx)
        bb = bt._batch
        bb.iterate(bt.seeds, 1)
        for i, name in enumerate(["fit_predict_cov", "factor", "normals", "gemm", "score", "kde"]):
            print("  batch of %d %-16s %.2f ms" % (B, name, bb.profile_stage(i, 3)), flush=True)
        bt.reset()
        t = time.time(); bt(); dt = time.time() - t
        print("batch of %d cold frames: %.3f s = %.2f frames/s, iterations %s" % (B, dt, B / dt, bt.timings["iters"]), flush=True)
    if "--oracle" in sys.argv:
        t = time.time(); orc.trace(init, grad, obs=warm, seed=3, sign_convention="harmonic", **kw)
        print("oracle frame (warm start) %.2f s" % (time.time() - t))


if __name__ == "__main__":
    main()
