"""The reference's DEFAULT kernel (kernel_options=(1, 3, 3): Matern-5/2, sigma_f = M // 6, l = edge_length // 2 -> any-rank factor) on the
headline's 500 x 500 edge: traces of one edge (for rocprofv3 --kernel-trace --stats) and wall-clock split.
usage: python tools/prof_default_kernel.py [edges=1] [reps=3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    import gaussian_process_edge_trace_amd as pkg
    from bench import README_KW, synth_image
    ctx = pkg._lib.Context(0)
    img, truth = synth_image(500, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    kw = dict(README_KW)
    kw["kernel_options"] = (1, 3, 3)
    one = pkg.GP_Edge_Tracing_Batch([init] * E, grad, [1 + 997 * e for e in range(E)], **kw, _ctx=ctx)
    one()
    for rep in range(reps):
        one.reset()
        t0 = time.time(); it = one.run_loop(); t1 = time.time(); one.finish(it); t2 = time.time()
        print("loop %.2f ms (%d..%d iterations), converged fit %.2f ms" % (1e3 * (t1 - t0), min(it), max(it), 1e3 * (t2 - t1)), flush=True)


if __name__ == "__main__":
    main()
