"""Warm start of the any-rank factor (option oj_warm): the factor stage on a Matern-5/2 batch three iterations into its trace
(covariance of the current observation set, rows of the previous iteration in the ring), with and without it.
usage: python tools/time_matern_warm.py [N] [B] [reps] [modes, e.g. 1 = the warm path only (for rocprofv3), default 01]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    modes = [int(c) for c in (sys.argv[4] if len(sys.argv) > 4 else "01")]
    import gaussian_process_edge_trace_amd as pkg
    from bench import synth_image
    L = pkg._lib
    ctx = L.Context(0)
    img, truth = synth_image(N, 5)
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1,
              N_samples=1000, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    seeds = [3 + e for e in range(B)]
    ref = None
    for warm in modes:
        old = L.set_option("oj_warm", warm)
        bt = pkg.GP_Edge_Tracing_Batch([init] * B, grad, seeds, **kw, _ctx=ctx)
        b = bt._batch
        b.iterate(seeds, 3)
        b.profile_stage(0, 1)        # fit + predict + covariance of the CURRENT observation set
        ms = b.profile_stage(1, reps)  # the factor of that covariance (previous rows: iteration 2's, in the ring)
        s = b.scalars()
        fac = np.array(b.read(L.BUF_FACTOR, 0))
        obs = b.read_obs_all()
        b.close()
        L.set_option("oj_warm", old)
        if ref is None:
            ref = (fac, obs)
        d = float(np.max(np.abs(fac - ref[0]))) / float(np.max(np.abs(ref[0])))
        print("oj_warm %d: N=%d B=%d factor %.2f ms per launch sequence (%d sweeps, rank %d, status %d); rows vs cold: max abs diff %.2e of the largest "
              "entry; observation sets after 3 iterations equal: %s" % (warm, N, B, ms, int(s.lml), s.rank, s.status, d,
                                                                         all(np.array_equal(a, c) for a, c in zip(obs, ref[1]))), flush=True)


if __name__ == "__main__":
    main()
