"""Summarise a rocprofv3 --kernel-trace of tools/single_edge_latency.py: per kernel name the launches, total device time and
the gaps between consecutive kernels -- i.e. what capturing the launch sequence in a hipGraph could save at most.
usage: python tools/single_edge_timeline.py <trace dir> [out.json]"""
import collections
import csv
import glob
import json
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete trace: from the last k_fit-family launch sequence back -- simply take the final third of the run
t_end = int(rows[-1]["End_Timestamp"])
names = [r["Kernel_Name"].split("(")[0].split("::")[-1] for r in rows]
# traces are separated by host work (> 200 us without a kernel): split there and keep the last full segment with a loop in it
segs, cur = [], [0]
for i in range(1, len(rows)):
    if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 200000:
        segs.append(cur)
        cur = []
    cur.append(i)
segs.append(cur)
out = {"segments": []}
for seg in segs[-6:]:
    if len(seg) < 20:
        continue
    dur = collections.Counter()
    cnt = collections.Counter()
    busy = 0
    for i in seg:
        d = int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
        dur[names[i]] += d
        cnt[names[i]] += 1
        busy += d
    span = int(rows[seg[-1]]["End_Timestamp"]) - int(rows[seg[0]]["Start_Timestamp"])
    gaps = [int(rows[seg[k + 1]]["Start_Timestamp"]) - int(rows[seg[k]]["End_Timestamp"]) for k in range(len(seg) - 1)]
    gaps_pos = [g for g in gaps if g > 0]
    top = sorted(dur.items(), key=lambda kv: -kv[1])[:8]
    s = {"launches": len(seg), "span_ms": span / 1e6, "kernel_busy_ms": busy / 1e6, "gap_total_ms": sum(gaps_pos) / 1e6,
         "gap_median_us": sorted(gaps_pos)[len(gaps_pos) // 2] / 1e3 if gaps_pos else 0.0,
         "top_kernels_ms": {k: [v / 1e6, cnt[k]] for k, v in top}}
    out["segments"].append(s)
    print("%4d launches: span %.2f ms, kernels busy %.2f ms, gaps %.2f ms (median %.1f us); top: %s" %
          (s["launches"], s["span_ms"], s["kernel_busy_ms"], s["gap_total_ms"], s["gap_median_us"],
           ", ".join("%s %.2f ms x%d" % (k, v / 1e6, cnt[k]) for k, v in top[:6])))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
