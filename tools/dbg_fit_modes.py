import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gaussian_process_edge_trace_amd as amd
L = amd._lib
ctx = L.Context(0)
g = np.load("/root/repo/tests/golden/trace_rbf500.npz"); st = np.load("/root/repo/tests/golden/stage_rbf500.npz")
from tests.test_oracle_vs_golden import CTOR
tr = amd.GP_Edge_Tracing(g["in_init"], st["ref_grad"], **CTOR["stage_rbf500"], _ctx=ctx)
n_iter = int(g["ref_n_iter"]); b = tr._batch
b.set_obs(0, g["ref_obs_%02d" % n_iter])
out = {}
for mode in (0, 1):
    L.set_option("fit_persistent", mode)
    out[mode] = b.final_fit_all([tr.seed + n_iter])
    print(mode, out[mode][2], out[mode][3], out[mode][4])
print("mean diff", np.abs(out[0][0] - out[1][0]).max(), "theta diff", np.abs(np.asarray(out[0][2]) - np.asarray(out[1][2])).max())
