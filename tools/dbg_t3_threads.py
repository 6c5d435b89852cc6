"""Does the oracle's trace (library sign convention) depend on the BLAS thread count of THIS host, and which variant is the
device's?  usage: python tools/dbg_t3_threads.py [img_seed] [rng_seed]   (default: the one T3 trace that left its fixture
row in round 4: image 3, seed 1000 + 997 * 179)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from threadpoolctl import threadpool_info, threadpool_limits  # noqa: E402
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from oracle import gpet_oracle as orc  # noqa: E402
from bench import README_KW  # noqa: E402

img_seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000 + 997 * 179
ctx = amd._lib.Context(0)
img, truth = orc.synth_sinusoid_image(500, img_seed)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
init = truth[[0, -1], :][:, [1, 0]]
print([(d.get("internal_api"), d.get("num_threads"), d.get("architecture")) for d in threadpool_info()], "cpus", os.cpu_count())
tr = amd.GP_Edge_Tracing(init, grad, seed=seed, **README_KW, _ctx=ctx)
et = tr()
print("device: %d iterations, MSE %.6f" % (tr._n_iter, amd.gpet_utils.trace_MSE(et, truth)))
for lim in (1, 2, 4, 8, None):
    with threadpool_limits(limits=lim):
        et_o, _, info = orc.trace(init, np.asarray(grad, dtype=np.float64), seed=seed, sign_convention="harmonic", blas_threads=None, **README_KW)
    print("oracle, %s BLAS thread(s): %d iterations, MSE %.6f, equal to the device: %s"
          % (lim, info["n_iter"], amd.gpet_utils.trace_MSE(et_o, truth), np.array_equal(et, et_o)))
