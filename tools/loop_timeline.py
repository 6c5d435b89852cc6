"""Timeline of one device loop (256 edges, no converged fits) from a rocprofv3 --kernel-trace CSV: per iteration,
kernel-busy time on the main stream vs wall time, and the gaps between consecutive kernels.
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/loop_timeline.py run
  python3 tools/loop_timeline.py report gpurun_out/tl"""
import csv
import glob
import os
import sys


def run():
    import numpy as np
    sys.path.insert(0, ".")
    import bench
    import gaussian_process_edge_trace_amd as pkg
    ctx = pkg._lib.Context(0)
    img, truth = bench.synth_image(500, 3)
    init = truth[[0, -1], :][:, [1, 0]]
    grad = pkg.gpet_utils.comp_grad_img(img, pkg.gpet_utils.kernel_builder((11, 5)), ctx=ctx)
    E = int(os.environ.get("GPET_TL_EDGES", "256"))
    tr = pkg.GP_Edge_Tracing_Batch([init] * E, grad, [1 + e for e in range(E)], **bench.README_KW, _ctx=ctx)
    for _ in range(3):
        tr.reset()
        tr.run_loop()
    ctx.sync()


def report(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    # the last loop: from the last third of k_fit launches on
    fits = [i for i, r in enumerate(rows) if "k_fit" in r["Kernel_Name"]]
    per = len(fits) // 3
    start = fits[-per]
    loop = rows[start:]
    main = [r for r in loop if "k_mt_normals" not in r["Kernel_Name"]]
    t0 = main[0]["s"]
    it_starts = [i for i, r in enumerate(main) if "k_fit" in r["Kernel_Name"]]
    print("iteration: wall us, busy us (main stream), largest gaps")
    for a, b in zip(it_starts[:14], it_starts[1:15]):
        seg = main[a:b]
        wall = main[b]["s"] - seg[0]["s"]
        busy = sum(r["e"] - r["s"] for r in seg)
        gaps = []
        for x, y in zip(seg, seg[1:] + [main[b]]):
            gaps.append((y["s"] - x["e"], x["Kernel_Name"].split("(")[0][-28:], y["Kernel_Name"].split("(")[0][-28:]))
        gaps.sort(reverse=True)
        print("%7.0f %7.0f  " % (wall / 1e3, busy / 1e3) + "; ".join("%.0f us %s->%s" % (g / 1e3, p, n) for g, p, n in gaps[:3]))
    seg = main[it_starts[2]:it_starts[3]]
    print("kernels of iteration 2:")
    for r in seg:
        print("  %8.1f +%7.1f us  %s" % ((r["s"] - seg[0]["s"]) / 1e3, (r["e"] - r["s"]) / 1e3, r["Kernel_Name"].split("(")[0][:60]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
