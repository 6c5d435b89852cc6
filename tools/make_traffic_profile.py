"""profiles/r01_pmc_traffic.json from two rocprofv3 PMC passes of tools/prof_stages.py:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/prof_stages.py 256 2
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/prof_stages.py 256 2
    python tools/make_traffic_profile.py 256 500

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half the
bytes of a streaming read (MI355X_MICROARCH.md, HBM section); calibrated here on two kernels with a
known byte count (k_score reads the 8 B/lane samples: 1.024 GB expected, 0.525 GB reported;
k_kde_normalise reads 256 MB of f32: 128.7 MB reported), WRITE_SIZE is exact (k_kde_normalise
writes 256 MB: 256.0 MB reported)."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last(path, name):
    f = glob.glob(os.path.join(ROOT, path, "*", "*_counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"].split("(")[0].replace("gpet::", "").replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    return {k: v[-1] for k, v in acc.items()}  # last launch = the 256-edge profile launch


def main():
    E, N = int(sys.argv[1]), int(sys.argv[2])
    fe, wr = last("gpurun_out/pmc_fetch", "FETCH_SIZE"), last("gpurun_out/pmc_write", "WRITE_SIZE")
    out = dict(edges=E, image=[N, N], source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/prof_stages.py",
               correction="HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024", kernels={})
    for k in sorted(fe):
        if k in wr:
            out["kernels"][k] = dict(fetch_size_kb=fe[k], write_size_kb=wr[k],
                                     hbm_bytes_per_launch=(2.0 * fe[k] + wr[k]) * 1024.0)
    # the LML kernel of the converged fits is not part of tools/prof_stages.py: its entry comes from two more PMC passes
    # over `tools/prof_final.py 256 0` (gpurun_out/pmc_lml_f, pmc_lml_w), averaged over the launches; keep it if present
    old_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(old_path):
        old = json.load(open(old_path)).get("kernels", {})
        if "k_lml" in old and "k_lml" not in out["kernels"]:
            out["kernels"]["k_lml"] = old["k_lml"]
    json.dump(out, open(old_path, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1)[:1500])


if __name__ == "__main__":
    main()
