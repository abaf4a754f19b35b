#!/bin/bash
# Frame-level HBM traffic, MEASURED ("report rocprof-measured HBM bytes next to the algorithmic figure"): one rocprofv3 --pmc
# FETCH_SIZE pass and one WRITE_SIZE pass (they do not fit one pass on gfx950: TCC has 4 counter slots, FETCH_SIZE takes 3) over the
# dependent c-main sequence, one frame at a time on one context (scripts/prof_dependent.py), summed over EVERY kernel of a frame.
# bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB -> bytes: the guide's gfx950 correction (FETCH_SIZE tallies 128-byte read requests at 64 bytes:
# exact for wide coalesced reads, an upper estimate for narrow gathers).
# Round 6: with nothing kept between frames (RGC_KNN_SEEDS=0 = rgc_set_knn_reuse(RGC_REUSE_NONE)), the workload bench.py's `value` times; a
# third argument `lists` measures the library's default on the unchanged synthetic map instead (-> frame_traffic_<tag>_lists.json).
#   usage (GPU box): bash scripts/frame_traffic.sh [frames, default 20] [tag, default cmain] [none|lists]   -> gpurun_out/frame_traffic_<tag>.json
K=${1:-20}; TAG=${2:-cmain}; REUSE=${3:-none}
if [ $REUSE = none ]; then export RGC_KNN_SEEDS=0; else unset RGC_KNN_SEEDS; fi
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ft_$TAG
rm -rf $O; mkdir -p $O
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $set -d $O/$set -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_dependent.py $K 0 $TAG > $O/$set.log 2>&1
done
cd $GRAFT_REPO_ROOT
K=$K TAG=$TAG REUSE=$REUSE python3 - <<'PY'
import csv, glob, json, collections, os, subprocess
root = os.environ["GRAFT_REPO_ROOT"]; K = int(os.environ["K"]); tag = os.environ["TAG"]
O = os.path.join(root, "gpurun_out", "ft_" + tag)
per = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rgck::", "")
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE": calls[name] += 1
frames = 2 * K + 2   # prof_dependent.py: one start-up frame per context, then the K frames twice
rows, tot = [], 0.0
for name, m in per.items():
    b = (2.0 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024.0 / frames
    tot += b
    rows.append({"kernel": name[:60], "launches_per_frame": round(calls[name] / frames, 2), "read_MB_per_frame": round(2.0 * m.get("FETCH_SIZE", 0.0) * 1024 / frames / 1e6, 3),
                 "written_MB_per_frame": round(m.get("WRITE_SIZE", 0.0) * 1024 / frames / 1e6, 3), "MB_per_frame": round(b / 1e6, 3)})
rows.sort(key=lambda r: -r["MB_per_frame"])
hc = os.path.join(root, ".head_commit")   # written beside the snapshot before the GPU call (there is no .git on the box)
commit = open(hc).read().strip() if os.path.exists(hc) else None
reuse = os.environ.get("REUSE", "none")
out = {"workload": tag + ": dependent sequence, one frame at a time, every kernel of a frame (scripts/prof_dependent.py), "
                   + ("nothing kept between frames (RGC_KNN_SEEDS=0)" if reuse == "none" else "the library's default: seeds + neighbour lists of the unchanged map"), "frames": frames,
       "bytes_per_frame_measured": int(tot), "formula": "(2 x FETCH_SIZE + WRITE_SIZE) KiB per kernel, summed (MI355X_MICROARCH.md: FETCH_SIZE x 2 on gfx950)",
       "commit": commit, "per_kernel": rows}
json.dump(out, open(os.path.join(root, "gpurun_out", "frame_traffic_" + tag + ("" if reuse == "none" else "_lists") + ".json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}))
for r in rows[:14]: print(r)
PY
rm -rf $O
unset RGC_KNN_SEEDS
