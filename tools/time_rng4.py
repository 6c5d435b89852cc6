"""The normals of a ring-filling launch at the bench's batch shape: one workgroup per stream (k_mt_normals) against the
register-resident generator with four streams per wave (k_mt_normals4).  usage: python tools/time_rng4.py [edges]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from bench import synth_image, README_KW  # noqa: E402
L = amd._lib
ctx = L.Context(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
img, truth = synth_image(500, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = [1 + 997 * e for e in range(E)]
tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
tr._batch.iterate(seeds, 7)
ring = tr._batch.info()["z_ring"]
for mode in (0, 1, 0, 1):
    L.set_option("rng4", mode)
    ms = tr._batch.profile_stage(2, 5)
    print("edges %d ring %d rng4=%d: %.3f ms per ring-filling launch = %.3f ms per iteration" % (E, ring, mode, ms, ms / ring), flush=True)
L.set_option("rng4", -1)
