#!/bin/bash
# Developer aid (GPU box): rocprofv3 kernel statistics of scripts/prof_dependent.py.   bash scripts/prof_dependent.sh [frames] [overlap 0|1] [cmain|c3|c5] [lazy margin]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_dep
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o dep --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_dependent.py ${1:-40} ${2:-0} ${3:-cmain} ${4:-0} > $O/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
frames = 2 * ${1:-40} + 2
tot = 0.0
for r in rows:
    per_frame = float(r["TotalDurationNs"]) / frames / 1e3
    tot += per_frame
    if per_frame > 0.5:
        print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), "%8.1f us avg" % (float(r["AverageNs"]) / 1e3), "%8.1f us/frame" % per_frame)
print("kernel time per frame: %.1f us" % tot)
PY
