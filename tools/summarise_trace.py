"""Average duration of the last `reps` launches of one kernel in a rocprofv3 --kernel-trace CSV."""
import csv, glob, json, os, sys


def main():
    d, name, reps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    source = sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --kernel-trace of tools/prof_dominant.py (the kernel alone, bench.py's mid-trace state)"
    f = sorted(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-reps:]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last]
    res = dict(kernel=name, launches=len(dur), average_ns=sum(dur) / len(dur), min_ns=min(dur), max_ns=max(dur),
               source=source,
               grid=[int(last[-1]["Grid_Size_" + a]) for a in "XYZ"], workgroup=[int(last[-1]["Workgroup_Size_" + a]) for a in "XYZ"],
               lds_bytes=int(last[-1]["LDS_Block_Size"]), vgpr=int(last[-1]["VGPR_Count"]))
    json.dump(res, open(out, "w"), indent=1)
    print(res)


if __name__ == "__main__":
    main()
