"""Runs the dominant loop kernel alone, at bench.py's mid-trace state, for a rocprofv3 --kernel-trace run:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_dom -- python3 tools/prof_dominant.py 256 122 20
    python tools/summarise_trace.py gpurun_out/prof_dom gpet::k_jacobi_ahead 20 profiles/r05_dominant_kernel.json

(stage id as in include/gpet_hip.h; the last `reps` launches of the trace are the profiled ones)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    stage = int(sys.argv[2]) if len(sys.argv) > 2 else 122
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    import gaussian_process_edge_trace_amd as pkg
    from bench import README_KW, synth_image
    ctx = pkg._lib.Context(0)
    img, truth = synth_image(500, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    tr = pkg.GP_Edge_Tracing_Batch([init] * E, grad, [1 + e for e in range(E)], **README_KW, _ctx=ctx)
    tr._batch.iterate(tr.seeds, 7)
    ms = tr._batch.profile_stage(stage, reps)
    print("stage %d: %.4f ms per launch (hipEvents, %d reps, %d edges)" % (stage, ms, reps, E))


if __name__ == "__main__":
    main()
