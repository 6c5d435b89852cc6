"""From a rocprofv3 --kernel-trace directory of tools/single_edge_latency.py: the kernels of ONE mid-trace iteration in start order,
with durations and the gap to the previous kernel's end.  usage: python tools/dump_iteration_kernels.py <trace dir> [anchor kernel] [which]"""
import csv
import glob
import re
import sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_fit"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(csv.DictReader(open(glob.glob(d + "/*/*_kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"^void\s+", "", r["Kernel_Name"].strip('"')).split("(")[0].replace("gpet::", "")
idx = [i for i, r in enumerate(rows) if name(r).startswith(anchor)]
a, b = idx[which], idx[which + 1]
prev_end = int(rows[a - 1]["End_Timestamp"])
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%6.1f gap  %7.1f us  q%-3s %s  grid %s wg %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name(r)[:60], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
    prev_end = max(prev_end, e)
print("iteration: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
