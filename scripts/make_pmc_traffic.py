"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over scripts/prof_frame.py.

    python scripts/make_pmc_traffic.py <fetch_dir> <write_dir> <out_json> <raw_json>

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and gfx950's FETCH_SIZE reports half of
wide (16 B per lane) reads (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Gather-shaped reads are uncalibrated: an upper estimate."""
import csv, glob, json, os, sys, collections
KIND = [("k_knn_sp<20, true>", "knn_cov_target"), ("k_knn_sp<20, false>", "knn_cov_source"), ("k_knn_coop<20, true>", "knn_coop_target"),
        ("k_knn_coop<20, false>", "knn_coop_source"), ("k_voxel_build", "voxel_build"), ("k_lm_step", "linearize"), ("k_fitness_lm", "fitness")]
def mean_per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void ", "").replace("rgck::", "")
            for pat, kind in KIND:
                if name.startswith(pat):
                    acc[kind].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fe, wr = mean_per_kernel(sys.argv[1], "FETCH_SIZE"), mean_per_kernel(sys.argv[2], "WRITE_SIZE")
raw = {k: {"FETCH_SIZE_KB": fe.get(k, 0.0), "WRITE_SIZE_KB": wr.get(k, 0.0)} for k in sorted(set(fe) | set(wr))}
out = {k: int(round((2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024)) for k, v in raw.items()}
json.dump(out, open(sys.argv[3], "w"), indent=1)
json.dump({"note": __doc__.strip().split("\n\n")[-1], "raw_KB": raw, "bytes_per_launch": out}, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
