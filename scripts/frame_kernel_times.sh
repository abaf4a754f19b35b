#!/bin/bash
# GPU time per kernel and frame of the timed workload -- the dependent c-main sequence on two contexts with nothing kept between frames
# (RGC_KNN_SEEDS=0 = rgc_set_knn_reuse(RGC_REUSE_NONE), what bench.py's `value` runs) -- from a rocprofv3 --kernel-trace --stats pass
# over scripts/prof_dependent.py.  -> gpurun_out/frame_kernel_times.json (bench.py's roofline_by_kernel reads profiles/r06_frame_kernel_times.json)
#   usage (GPU box): bash scripts/frame_kernel_times.sh [frames, default 40] [overlap 0|1, default 1] [tag cmain|c3|c5]
K=${1:-40}; OV=${2:-1}; TAG=${3:-cmain}
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/fkt; rm -rf $O; mkdir -p $O
RGC_KNN_SEEDS=0 rocprofv3 --kernel-trace --stats -d $O -o dep --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_dependent.py $K $OV $TAG > $O/run.log 2>&1
cd $GRAFT_REPO_ROOT
K=$K OV=$OV TAG=$TAG python3 - <<'PY'
import csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"]; K = int(os.environ["K"])
f = glob.glob(os.path.join(root, "gpurun_out", "fkt", "**", "*kernel_stats.csv"), recursive=True)[0]
frames = 2 * K + 2   # prof_dependent.py: one start-up frame per context, then the K frames twice
rows, tot = [], 0.0
for r in csv.DictReader(open(f)):
    name = r["Name"].split("(")[0].replace("void ", "").replace("rgck::", "")
    us = float(r["TotalDurationNs"]) / frames / 1e3
    tot += us
    rows.append({"kernel": name[:60], "launches_per_frame": round(int(r["Calls"]) / frames, 2), "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "us_per_frame": round(us, 2)})
rows.sort(key=lambda r: -r["us_per_frame"])
hc = os.path.join(root, ".head_commit")
out = {"workload": os.environ["TAG"] + ": dependent sequence, " + ("two contexts" if os.environ["OV"] == "1" else "one frame at a time") + ", nothing kept between frames "
                   "(RGC_KNN_SEEDS=0), rocprofv3 --kernel-trace --stats over scripts/prof_dependent.py", "frames": frames,
       "kernel_us_per_frame_total": round(tot, 1), "commit": open(hc).read().strip() if os.path.exists(hc) else None, "per_kernel": rows}
json.dump(out, open(os.path.join(root, "gpurun_out", "frame_kernel_times.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}))
for r in rows[:12]: print(r)
PY
rm -rf $O
