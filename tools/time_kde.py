"""k_kde_fused (stage 151) and the curve-KDE stage on the bench batch, on traces with INDEPENDENT seeds (997 apart, as bench.py
draws them) and with consecutive ones, at several points of the trace.  usage: python tools/time_kde.py [edges]"""
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
for name, seeds in (("spaced", [1 + 997 * e for e in range(E)]), ("consecutive", list(range(1, E + 1)))):
    tr = amd.GP_Edge_Tracing_Batch([init] * E, grad, seeds, **README_KW, _ctx=ctx)
    done = 0
    for upto in (3, 7, 11):
        tr._batch.iterate(seeds, upto - done)
        done = upto
        b = tr._batch
        for k in (130, 140, 141, 150):  # (the state the loop's KDE sees: this iteration's samples, costs, kept curves, weights)
            b.profile_stage(k, 1)
        t = {}
        for form in (0, 1, 0, 1):
            L.set_option("kde_form", form)
            t.setdefault(form, []).append(b.profile_stage(151, 10))
        L.set_option("kde_form", 1)
        print("%s seeds, %2d iterations in: k_kde_fused form 0 (round 3) %s | form 1 (per-wave columns) %s ms"
              % (name, upto, *[" ".join("%.3f" % v for v in t[f]) for f in (0, 1)]), flush=True)
    tr._batch.close()
