"""Where a device trace leaves the oracle's (same sign convention): first iteration whose observation set differs, and what
differs in it -- costs, kept curves, KDE, pixels.  usage: python tools/dbg_t3_seed.py [img_seed] [rng_seed] [orc]
(orc: both on the ORACLE's gradient image rounded to float32 -- the fixture's -- instead of the library's own)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import gaussian_process_edge_trace_amd as amd  # noqa: E402
from oracle import gpet_oracle as orc  # noqa: E402
from bench import README_KW  # noqa: E402

img_seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000 + 997 * 179
L = amd._lib
ctx = L.Context(0)
img, truth = orc.synth_sinusoid_image(500, img_seed)
grad = amd.gpet_utils.comp_grad_img(img, amd.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
if len(sys.argv) > 3 and sys.argv[3] == "orc":
    g2 = orc.comp_grad_img(img, orc.kernel_builder((11, 5))).astype(np.float32)
    print("gradient images: library vs oracle rounded to float32: %d of %d pixels differ, max abs %.3g"
          % (int(np.sum(np.asarray(grad) != g2)), g2.size, float(np.max(np.abs(np.asarray(grad, dtype=np.float64) - g2)))))
    grad = g2
init = truth[[0, -1], :][:, [1, 0]]
rec = []
et_o, _, info = orc.trace(init, grad, seed=seed, record=rec, sign_convention="harmonic", **README_KW)
tr = amd.GP_Edge_Tracing(init, grad, seed=seed, **README_KW, _ctx=ctx)
et, (all_samples, all_obs, curves) = tr(return_lines=True)
print("iterations: device %d, oracle %d; traces equal: %s" % (tr._n_iter, info["n_iter"], np.array_equal(et, et_o)))
p = orc.resolve_params(init, grad, seed=seed, **README_KW)
grad64 = orc.normalise(grad, (0, 1), np.float64)
for i, r in enumerate(rec):
    if i + 1 >= len(all_obs):
        print("device stopped before iteration", i)
        break
    same = np.array_equal(all_obs[i + 1], r["obs_out"])
    if same:
        continue
    print("first difference: iteration %d: device %d observations, oracle %d" % (i, len(all_obs[i + 1]), len(r["obs_out"])))
    a = {tuple(v) for v in np.asarray(all_obs[i + 1]).tolist()}
    o = {tuple(v) for v in np.asarray(r["obs_out"]).tolist()}
    print("  only device:", sorted(a - o)[:10], " only oracle:", sorted(o - a)[:10])
    # the iteration's samples on the device, scored by the oracle's cost function
    Y = np.asarray(all_samples[i])  # (N, S)
    costs_dev_samples = orc.costs_batch(grad64, p["x_grid"], Y)
    co = np.asarray(r["costs"])
    nk = p["N_keep"]
    od, oo = np.argsort(costs_dev_samples, kind="stable")[:nk], np.asarray(r["best_idxs"])
    print("  oracle costs of the device's samples vs the oracle's own costs: max rel diff %.3g" % np.max(np.abs(costs_dev_samples / co - 1)))
    print("  kept sets equal: %s; as ordered lists: %s" % (set(od.tolist()) == set(oo.tolist()), np.array_equal(od, oo)))
    if set(od.tolist()) != set(oo.tolist()):
        x = sorted(set(od.tolist()) ^ set(oo.tolist()))
        print("  curves in one kept set only:", x, "their costs (device samples):", costs_dev_samples[x], "(oracle):", co[x])
        s = np.sort(co)
        print("  oracle costs around the cut: %r" % s[nk - 2:nk + 2])
    break
else:
    print("all iterations equal")
