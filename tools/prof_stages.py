"""Runs each per-iteration stage a few times on a mid-trace batch state (for rocprofv3 runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    import gaussian_process_edge_trace_amd as pkg
    from bench import README_KW, synth_image, STAGES
    ctx = pkg._lib.Context(0)
    img, truth = synth_image(500, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    tr = pkg.GP_Edge_Tracing_Batch([init] * E, grad, list(range(1, E + 1)), **README_KW, _ctx=ctx)
    tr._batch.iterate(tr.seeds, 7)
    s = tr._batch.scalars(0)
    print("mid-trace state: n=%d rank=%d jacobi sweeps=%d" % (s.n, s.rank, int(s.lml)), flush=True)
    for i, name in enumerate(STAGES):
        ms = tr._batch.profile_stage(i, reps)
        print("%-16s %.3f ms" % (name, ms), flush=True)


if __name__ == "__main__":
    main()
