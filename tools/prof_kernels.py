"""Per-kernel durations of one mid-trace iteration (run under rocprofv3 --kernel-trace)."""
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
    tr._batch.iterate(tr.seeds, 7)
    for i in range(6):
        tr._batch.profile_stage(i, 3)


if __name__ == "__main__":
    main()
