"""Sample GEMM, scorer and curve KDE of the bench batch at its mid-trace state with f64 and with f32 samples."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_edge_trace_amd as amd
import bench
from bench import synth_image, README_KW
L = amd._lib
ctx = L.Context(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = 500
img, truth = synth_image(N, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = list(range(1, E + 1))
ids = dict(bench.KERNEL_IDS_STRUCT); ids.update(bench.KERNEL_IDS_COMMON)
want = [k for k, v in sorted(ids.items()) if any(t in v for t in ("gemm", "score", "kde", "pix", "topk"))]
for dt in ("f64", "f32", "f64"):
    tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx, sample_dtype=dt)
    tr._batch.iterate(seeds, 7)
    print(dt, "  ".join("%s %.3f" % (ids[k], tr._batch.profile_stage(k, 20)) for k in want), flush=True)
    tr._batch.close()
