"""Accuracy of the objective kernels against the oracle on the fixtures' final training sets (40 random theta each)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import gaussian_process_edge_trace_amd as amd
from tests import final_fit_inputs as ff
from oracle import gpet_oracle as orc
from tests.test_oracle_vs_golden import CTOR
L = amd._lib
ctx = L.Context(0)
for name, stage in [("trace_rbf500", "stage_rbf500"), ("trace_mat128", "stage_mat128")]:
    g = dict(np.load("tests/golden/%s.npz" % name))
    grad = np.load("tests/golden/%s.npz" % stage)["ref_grad"]
    tr = amd.GP_Edge_Tracing(g["in_init"], grad, **CTOR[stage], _ctx=ctx)
    last = max(int(k[8:]) for k in g if k.startswith("ref_obs_"))
    pr = ff.prepare(tr.init, g["ref_obs_%02d" % last], tr.x_grid, tr.fix_endpoints)
    b = tr._batch
    b.final_set_training(0, pr["xs"], pr["yt"], pr["w"])
    rng = np.random.default_rng(0)
    th = ff.BOUNDS[:, 0] + (ff.BOUNDS[:, 1] - ff.BOUNDS[:, 0]) * rng.uniform(size=(40, 3))
    th[:, 2] = np.log(rng.uniform(1e-4, 1.0, size=40))
    th[0] = np.log([5.0, 5.0, 1.0])
    ref = [orc.lml_and_grad(t, pr["xs"], pr["yt"], pr["w"], tr.kernel_type, tr.kernel_nu) for t in th]
    for mode in (0, 1):
        L.set_option("lml_mfma", mode)
        f, gr = b.lml_batch(np.zeros(40, dtype=np.int32), th)
        ef = [abs(f[i] + ref[i][0]) / (1 + abs(ref[i][0])) for i in range(40) if np.isfinite(ref[i][0])]
        eg = [np.abs(gr[i] + ref[i][1]).max() / (1 + np.abs(ref[i][1]).max()) for i in range(40) if np.isfinite(ref[i][0])]
        k = int(np.argmax(ef))
        print("%s n=%d mode %d: max rel err f %.2e (theta %s, cond-ish c/nl %.1e), g %.2e" %
              (name, pr["xs"].size, mode, max(ef), np.round(np.exp(th[k]), 5), np.exp(th[k][0] - th[k][2]), max(eg)))
    L.set_option("lml_mfma", 1)
