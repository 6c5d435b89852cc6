"""GPU idle time inside the bench's timed steps, from a rocprofv3 --kernel-trace of bench.py.
usage: python tools/bench_timeline.py <trace dir> [out.json] [edges]
Union of all kernel intervals (all streams) over the last third of the run: busy time, idle time, and the idle gaps by
the kernel that follows them -- what the host or a dependency keeps the GPU waiting for."""
import collections
import csv
import glob
import json
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1]) for r in rows)
# the headline steps: the k_struct_H launches over 1024 edges; the timed ones are the last `steps` of steps + warmup
E = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
hs = sorted(int(r["Start_Timestamp"]) for r in rows if "k_struct_H" in r["Kernel_Name"] and int(r["Grid_Size_Y"]) == E)
# split into runs separated by > 50 ms (the secondaries also run 1024-edge batches: ctor_included, no_pipeline)
runs, cur = [], [hs[0]]
for t in hs[1:]:
    if t - cur[-1] > 50e6:
        runs.append(cur)
        cur = []
    cur.append(t)
runs.append(cur)
run = max(runs, key=len)  # the headline: steps + warmup batches back to back
lo, hi = run[len(run) // 3], run[-1]
iv = [x for x in iv if lo <= x[0] <= hi]
busy, idle, cur_end = 0, 0, iv[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
big = []
for s, e, n in iv:
    if s > cur_end:
        g = s - cur_end
        idle += g
        gaps[n][0] += g
        gaps[n][1] += 1
        if g > 200000:
            big.append((g / 1e3, n))
        cur_end = s
    if e > cur_end:
        busy += e - cur_end
        cur_end = e
out = dict(window_ms=(iv[-1][1] - iv[0][0]) / 1e6, busy_ms=busy / 1e6, idle_ms=idle / 1e6,
           idle_before_kernel_ms={k: [round(v[0] / 1e6, 3), v[1]] for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]},
           gaps_over_200us=sorted(big, reverse=True)[:20])
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
