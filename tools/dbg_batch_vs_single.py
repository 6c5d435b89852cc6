"""One edge of a batch against the same edge traced alone: is the difference reproducible, and which small-batch code path
(rotation-log Jacobi, one-workgroup-per-problem fits, chunked generator, split GEMM / scorer) makes it?
usage: python tools/dbg_batch_vs_single.py [img_seed] [n_seeds] [index]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from oracle import gpet_oracle as orc  # noqa: E402
from bench import README_KW  # noqa: E402

img_seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 240
idx = int(sys.argv[3]) if len(sys.argv) > 3 else 179
L = amd._lib
ctx = L.Context(0)
img, truth = orc.synth_sinusoid_image(500, img_seed)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
init = truth[[0, -1], :][:, [1, 0]]
seeds = [1000 + 997 * k for k in range(n)]


def batch(sd):
    bt = amd.GP_Edge_Tracing_Batch([init] * len(sd), np.asarray(grad, dtype=np.float32), sd, **README_KW, _ctx=ctx)
    tr = bt()
    it = list(bt.timings["iters"])
    obs = bt._batch.read_obs_all()
    bt._batch.close()
    return [np.asarray(t) for t in tr], it, obs


t1, i1, o1 = batch(seeds)
t2, i2, o2 = batch(seeds)
print("batch of %d twice: iterations equal %s, traces equal %s" % (n, i1 == i2, all(np.array_equal(a, b) for a, b in zip(t1, t2))))
ts, is_, os_ = batch([seeds[idx]])
print("edge %d (seed %d): batch iterations %d, alone %d; traces equal %s; final observation sets equal %s"
      % (idx, seeds[idx], i1[idx], is_[0], np.array_equal(t1[idx], ts[0]), np.array_equal(o1[idx], os_[0])))
bad = []
for k in range(n):
    tk, ik, ok = batch([seeds[k]]) if k in (idx,) else (None, None, None)
for name, opts in (("jlog_max_b=0", {"jlog_max_b": 0}), ("fit_persistent=0", {"fit_persistent": 0}), ("rng_chunked=0", {"rng_chunked": 0}),
                   ("all three", {"jlog_max_b": 0, "fit_persistent": 0, "rng_chunked": 0})):
    old = {k: L.set_option(k, v) for k, v in opts.items()}
    try:
        tk, ik, ok = batch([seeds[idx]])
    finally:
        for k, v in old.items():
            L.set_option(k, v)
    print("  alone with %-18s iterations %d, equals the batch's trace: %s, equals the default single-edge trace: %s"
          % (name + ":", ik[0], np.array_equal(tk[0], t1[idx]), np.array_equal(tk[0], ts[0])))
# a smaller batch around it
for m in (2, 8, 33, 65):
    lo = max(0, idx - m + 1)
    sd = seeds[lo:idx + 1]
    tk, ik, ok = batch(sd)
    print("  in a batch of %3d (last edge): iterations %d, equals the big batch's trace %s, the single-edge trace %s"
          % (len(sd), ik[-1], np.array_equal(tk[-1], t1[idx]), np.array_equal(tk[-1], ts[0])))
