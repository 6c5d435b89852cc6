"""Sample GEMM + scorer of the bench batch at its mid-trace state: separate kernels (stages 130, 140) against the fused
form of the device loop (131 = samples scored out of the accumulators + combine, 132 = the kept rows), and the loop
itself with and without it."""
import sys, os, time
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
tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
b = tr._batch
b.iterate(seeds, 7)
for rep in range(2):
    t = {k: b.profile_stage(k, 20) for k in (130, 140, 141, 132)}
    f = {}
    for variant in (1, 2):  # 1: column-tile-stationary (k_sample_score), 2: curve-stationary (k_sample_score2)
        L.set_option("fused_score", variant)
        f[variant] = b.profile_stage(131, 20)
    L.set_option("fused_score", 0)
    print("E=%d  gemm %.3f + score %.3f = %.3f ms | fused (column tiles) %.3f, (curve blocks) %.3f + keep rows %.3f = %.3f / %.3f ms | topk %.3f"
          % (E, t[130], t[140], t[130] + t[140], f[1], f[2], t[132], f[1] + t[132], f[2] + t[132], t[141]), flush=True)
for fused in (2, 0, 1, 2, 0):
    L.set_option("fused_score", fused)
    tr.reset()
    ctx.sync()
    t0 = time.time()
    it = tr.run_loop()
    ctx.sync()
    print("fused_score=%d: loop %.1f ms (%d..%d iterations)" % (fused, 1e3 * (time.time() - t0), min(it), max(it)), flush=True)
b.close()
