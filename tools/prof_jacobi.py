"""Ad-hoc: eigen-solver kernel time and sweep count at several points of a trace (structured loop path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    import gaussian_process_edge_trace_amd as pkg
    from bench import README_KW, synth_image
    ctx = pkg._lib.Context(0)
    img, truth = synth_image(500, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    tr = pkg.GP_Edge_Tracing_Batch([init] * E, grad, list(range(1, E + 1)), **README_KW, _ctx=ctx)
    b = tr._batch
    print("info", b.info())
    done = 0
    for it in (1, 3, 3, 5, 5):
        b.iterate(tr.seeds, it)
        done += it
        ms = b.profile_stage(122, 10)
        sc = b.scalars(0)
        sw = [int(b.scalars(e).lml) for e in range(0, E, max(1, E // 8))]
        print("after %2d iterations: n=%3d rank=%d eig kernel %.3f ms, sweeps %s" % (done, sc.n, sc.rank, ms, sw), flush=True)


if __name__ == "__main__":
    main()
