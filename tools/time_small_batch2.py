"""One batch object alone (reset -> device loop -> converged fits): median ms per step for option settings given on the command line.
usage: python tools/time_small_batch2.py <edges> <reps> [name=value,name=value ...]   (each further argument = one configuration)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in (sys.argv[3:] or [""])]
    import gaussian_process_edge_trace_amd as amd
    from bench import synth_image, README_KW
    L = amd._lib
    img, truth = synth_image(500, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    seeds = [1 + 997 * e for e in range(E)]
    ref = None
    rounds = int(os.environ.get("GPET_AB_ROUNDS", "1"))  # > 1: the configurations interleaved, a fresh object every time
    for opts in cfgs * rounds:
        old = {k: L.set_option(k, v) for k, v in opts.items()}
        try:
            ctx = L.Context(0)
            grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
            tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx, rng=os.environ.get("GPET_AB_RNG") or None)
            tl, tf = [], []
            for _ in range(reps + 1):
                ctx.sync()
                t0 = time.time()
                tr.reset()
                it = tr.run_loop()
                ctx.sync()
                t1 = time.time()
                out = tr.finish(it)
                ctx.sync()
                tl.append(1e3 * (t1 - t0))
                tf.append(1e3 * (time.time() - t1))
            same = "" if ref is None else (" traces equal the first configuration's: %s" % all(np.array_equal(a, b) for a, b in zip(out, ref)))
            if ref is None:
                ref = out
            print("%d edges alone, %-28s loop %.2f ms (%d..%d iterations) + converged fits %.2f ms = %.2f ms per step%s"
                  % (E, (",".join("%s=%d" % kv for kv in opts.items()) or "defaults") + ":", np.median(tl[1:]), min(it), max(it),
                     np.median(tf[1:]), np.median(tl[1:]) + np.median(tf[1:]), same), flush=True)
            tr._batch.close()
        finally:
            for k, v in old.items():
                L.set_option(k, v)


if __name__ == "__main__":
    main()
