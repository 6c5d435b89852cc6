"""Eigen stage (pre-rotation + Jacobi [+ rotation-log pass]) of a batch at the bench's mid-trace state: the form that updates W inside
the rounds against the rotation-log form with the log pass sharing the log through LDS (k_jacobi_wpass_lds), per batch size.
One process per configuration (jlog_max_b is read when a batch is created).  usage: python tools/time_jacobi_logform.py <edges>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gaussian_process_edge_trace_amd as amd
from bench import synth_image, README_KW
L = amd._lib
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = L.Context(0)
img, truth = synth_image(500, 3)
init = truth[[0, -1], :][:, [1, 0]]
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
seeds = [1 + 997 * e for e in range(E)]
tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
b = tr._batch
b.iterate(seeds, 7)
b.profile_stage(120, 1)  # the fit and H of the NEXT iteration (its observation set is on the device): the Jacobi below then has real work,
b.profile_stage(121, 1)  # warm-started from the last iteration's eigenvectors like in the loop
print("%d edges, jlog_max_b=%d wpass_lds=%d: eigen stage %.3f ms (sweeps %d)" % (E, L.get_option("jlog_max_b"), L.get_option("wpass_lds"),
                                                                                  b.profile_stage(122, 20), int(b.scalars(0).lml)), flush=True)
