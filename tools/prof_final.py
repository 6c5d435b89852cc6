"""The converged fits of one batch alone (device-resident L-BFGS-B + batched LML objective), for timing and for
rocprofv3 runs.  usage: python tools/prof_final.py [edges] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else 3
    import gaussian_process_edge_trace_amd as pkg
    from bench import README_KW, synth_image
    ctx = pkg._lib.Context(0)
    img, truth = synth_image(500, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    tr = pkg.GP_Edge_Tracing_Batch([init] * E, grad, list(range(1, E + 1)), **README_KW, _ctx=ctx)
    iters = tr.run_loop()
    b = tr._batch
    for rep in range(reps):
        b.lml_stats(reset=True)
        t0 = time.time()
        tr.finish(iters)
        dt = time.time() - t0
        st = b.lml_stats()
        print("E=%d converged fits %.1f ms wall (%d rounds; objective kernels %.1f ms, %d evaluations in %d launches)"
              % (E, 1e3 * dt, tr._fit_rounds, st["kernel_ms"], st["evaluations"], st["launches"]), flush=True)


if __name__ == "__main__":
    main()
