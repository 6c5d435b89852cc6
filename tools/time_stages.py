"""Per-kernel device time of the bench batch at its mid-trace state (gpet_profile_stage ids of bench.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
tr._batch.iterate(seeds, 7)
ids = dict(bench.KERNEL_IDS_STRUCT); ids.update(bench.KERNEL_IDS_COMMON)
for kid, name in sorted(ids.items()):
    print("%-24s %8.3f ms" % (name, tr._batch.profile_stage(kid, 20)), flush=True)
for i, name in enumerate(bench.STAGES):
    print("stage %-18s %8.3f ms" % (name, tr._batch.profile_stage(i, 20)), flush=True)
