"""profiles/<round>_pmc_traffic.json from two rocprofv3 PMC passes of tools/prof_stages.py:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/prof_stages.py 1024 2
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/prof_stages.py 1024 2
    (the same two passes over tools/prof_final.py 1024 0 into gpurun_out/pmc_lml_f, pmc_lml_w)
    python tools/make_traffic_profile.py 1024 500        -- all of it: tools/refresh_profiles.sh

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
    return {k: v[-1] for k, v in acc.items()}  # last launch = the full-batch profile launch


def mean_of(path, name, kernel):
    f = glob.glob(os.path.join(ROOT, path, "*", "*_counter_collection.csv"))[0]
    # (exact kernel: "gpet::k_lml16(" does not pick up k_lml16_fit or the register-tile kernels k_lml / k_lml2)
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
         if r["Counter_Name"] == name and r["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0] == kernel]
    return sum(v) / len(v), len(v)


def main():
    E, N = int(sys.argv[1]), int(sys.argv[2])
    R = sys.argv[3] if len(sys.argv) > 3 else "r05"
    fe, wr = last("gpurun_out/pmc_fetch", "FETCH_SIZE"), last("gpurun_out/pmc_write", "WRITE_SIZE")
    out = dict(edges=E, image=[N, N], source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/prof_stages.py",
               correction="HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024", kernels={})
    for k in sorted(fe):
        if k in wr:
            out["kernels"][k] = dict(fetch_size_kb=fe[k], write_size_kb=wr[k],
                                     hbm_bytes_per_launch=(2.0 * fe[k] + wr[k]) * 1024.0)
    # the LML kernel of the converged fits is not part of tools/prof_stages.py: its entry comes from two more PMC passes
    # over `tools/prof_final.py E 0` (gpurun_out/pmc_lml_f, pmc_lml_w), averaged over all its launches
    old_path = os.path.join(ROOT, "profiles", R + "_pmc_traffic.json")
    if glob.glob(os.path.join(ROOT, "gpurun_out/pmc_lml_f", "*", "*_counter_collection.csv")):
        lf, lw = mean_of("gpurun_out/pmc_lml_f", "FETCH_SIZE", "gpet::k_lml16"), mean_of("gpurun_out/pmc_lml_w", "WRITE_SIZE", "gpet::k_lml16")
        out["kernels"]["k_lml"] = dict(fetch_size_kb=lf[0], write_size_kb=lw[0], launches=lf[1],
                                       hbm_bytes_per_launch=(2.0 * lf[0] + lw[0]) * 1024.0,
                                       source="tools/prof_final.py %d 0: mean over the LML launches of one batch's converged fits" % E)
    elif os.path.exists(old_path):
        old = json.load(open(old_path))
        if old.get("edges") == E and "k_lml" in old.get("kernels", {}):
            out["kernels"]["k_lml"] = old["kernels"]["k_lml"]
    json.dump(out, open(old_path, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1)[:1500])


if __name__ == "__main__":
    main()
