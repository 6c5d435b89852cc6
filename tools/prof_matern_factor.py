"""The full-rank factor alone (BASELINE config 5's frame shape: 1024 columns, Matern-5/2) for rocprofv3 / timing.
usage: python tools/prof_matern_factor.py [N] [B] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    import gaussian_process_edge_trace_amd as pkg
    from bench import synth_image
    ctx = pkg._lib.Context(0)
    img, truth = synth_image(N, 5)
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    init = truth[[0, -1], :][:, [1, 0]]
    warm = truth[16:-16:16][:, [1, 0]].astype(np.int64)
    kw = dict(kernel_options={'kernel': 'Matern', 'nu': 2.5, 'sigma_f': 0.15 * N, 'length_scale': 0.04 * N}, noise_y=1,
              N_samples=1000, score_thresh=1, delta_x=8, keep_ratio=0.1, pixel_thresh=5, fix_endpoints=True)
    bt = pkg.GP_Edge_Tracing_Batch([init] * B, grad, [3 + e for e in range(B)], **kw, _ctx=ctx)
    b = bt._batch
    for e in range(B):
        b.set_obs(e, warm)
    b.fit_predict(True)
    b.factor()
    s = b.scalars()
    t = time.time()
    ms = b.profile_stage(1, reps)
    print("N=%d B=%d factor %.2f ms per launch sequence (%d sweeps, rank %d); host wall %.1f ms per rep"
          % (N, B, ms, int(s.lml), s.rank, (time.time() - t) * 1e3 / reps))


if __name__ == "__main__":
    main()
