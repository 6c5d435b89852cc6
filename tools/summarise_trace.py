"""Durations of ONE kernel in a rocprofv3 --kernel-trace CSV: the kernel name is matched EXACTLY up to its argument list
(`gpet::k_lml16` does not pick up `gpet::k_lml16_fit` or `gpet::k_lml`), template instantiations are reported one by one, and
grid / workgroup / LDS / VGPR figures are those of the MODAL launch shape, not of whatever launch came last.
usage: python tools/summarise_trace.py <trace dir> <kernel name, e.g. gpet::k_lml16> <last N launches (0 = all)> <out.json> [source note]"""
import collections
import csv
import glob
import json
import os
import re
import sys


def base_name(full):
    """'void gpet::k_x<2, true>(gpet::EdgeDev*, int)' -> ('gpet::k_x', '<2, true>')"""
    s = full.strip().strip('"')
    s = re.sub(r"^void\s+", "", s)
    s = s.split("(")[0]
    m = re.match(r"^([^<]+)(<.*>)?$", s)
    return (m.group(1), m.group(2) or "") if m else (s, "")


def main():
    d, name, reps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    source = sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --kernel-trace of tools/prof_dominant.py (the kernel alone, bench.py's mid-trace state)"
    f = sorted(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if base_name(r["Kernel_Name"])[0] == name]
    if not rows:
        raise SystemExit("no launch of %r in %s (exact match on the name in front of '<' / '(')" % (name, f))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if reps > 0:
        rows = rows[-reps:]
    variants = collections.OrderedDict()
    for r in rows:
        variants.setdefault(base_name(r["Kernel_Name"])[1], []).append(r)

    def stats(rs):
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
        shape = collections.Counter((tuple(int(r["Grid_Size_" + a]) for a in "XYZ"), tuple(int(r["Workgroup_Size_" + a]) for a in "XYZ"),
                                     int(r["LDS_Block_Size"]), int(r["VGPR_Count"])) for r in rs)
        (grid, wg, lds, vgpr), n_modal = shape.most_common(1)[0]
        grids = sorted({tuple(int(r["Grid_Size_" + a]) for a in "XYZ") for r in rs})
        return dict(launches=len(dur), average_ns=sum(dur) / len(dur), min_ns=min(dur), max_ns=max(dur),
                    modal_launch=dict(grid=list(grid), workgroup=list(wg), lds_bytes=lds, vgpr=vgpr, launches_of_this_shape=n_modal),
                    grid_x_range=[grids[0][0], grids[-1][0]])
    res = dict(kernel=name, source=source, **stats(rows))
    if len(variants) > 1 or "" not in variants:
        res["variants"] = {name + v: stats(rs) for v, rs in variants.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
