"""Where the time of one small batch of 500x500 edges goes: device loop vs converged fit (host wall clock).
usage: python tools/single_edge_latency.py [edges=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import gaussian_process_edge_trace_amd as pkg
    from bench import README_KW, synth_image
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    ctx = pkg._lib.Context(0)
    img, truth = synth_image(500, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    one = pkg.GP_Edge_Tracing_Batch([init] * E, grad, [1 + 997 * e for e in range(E)], **README_KW, _ctx=ctx)
    one()
    for rep in range(3):
        one.reset()
        t0 = time.time(); it = one.run_loop(); t1 = time.time(); one.finish(it); t2 = time.time()
        st = one._batch.lml_stats(reset=True)
        print("loop %.2f ms (%d iterations), converged fit %.2f ms (%d rounds, objective kernels %.2f ms)"
              % (1e3 * (t1 - t0), max(it), 1e3 * (t2 - t1), one._fit_rounds, st["kernel_ms"]), flush=True)
    b = one._batch
    one.reset()
    t0 = time.time(); b.iterate(one.seeds, 1); t1 = time.time()
    print("one iteration alone (enqueue + sync): %.2f ms" % (1e3 * (t1 - t0)))
    t0 = time.time(); b.iterate(one.seeds, 8); t1 = time.time()
    print("eight iterations in one call: %.2f ms" % (1e3 * (t1 - t0)))


if __name__ == "__main__":
    main()
