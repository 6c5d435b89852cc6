"""One batch object ALONE on the GPU (reset -> device loop -> converged fits), the literal per-GPU share of BASELINE config 4
(256 edges / 8 GPUs = 32): median ms per step and its split, for a few option settings.
usage: python tools/time_small_batch.py [edges] [reps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    import gaussian_process_edge_trace_amd as amd
    from bench import synth_image, README_KW
    L = amd._lib
    img, truth = synth_image(500, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    seeds = [1 + 997 * e for e in range(E)]
    for name, opts in (("defaults", {}), ("jlog_max_b=64", {"jlog_max_b": 64}), ("jlog_max_b=0", {"jlog_max_b": 0}),
                       ("rng4=1", {"rng4": 1}), ("rng_chunked=1", {"rng_chunked": 1}), ("fit_persistent=0", {"fit_persistent": 0}),
                       ("rng_inline=2", {"rng_inline": 2}), ("rng_inline=1", {"rng_inline": 1}), ("rng_lookahead=1", {"rng_lookahead": 1}),
                       ("rng_lookahead=4", {"rng_lookahead": 4})):
        old = {k: L.set_option(k, v) for k, v in opts.items()}
        try:
            ctx = L.Context(0)
            grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
            tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
            tl, tf = [], []
            for _ in range(reps + 1):
                ctx.sync()
                t0 = time.time()
                tr.reset()
                it = tr.run_loop()
                ctx.sync()
                t1 = time.time()
                tr.finish(it)
                ctx.sync()
                tl.append(1e3 * (t1 - t0))
                tf.append(1e3 * (time.time() - t1))
            print("%d edges alone, %-18s loop %.2f ms (%d..%d iterations) + converged fits %.2f ms = %.2f ms per step"
                  % (E, name + ":", np.median(tl[1:]), min(it), max(it), np.median(tf[1:]), np.median(tl[1:]) + np.median(tf[1:])), flush=True)
            tr._batch.close()
        finally:
            for k, v in old.items():
                L.set_option(k, v)


if __name__ == "__main__":
    main()
