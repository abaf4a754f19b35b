#!/bin/bash
# Developer aid (GPU box): per-kernel averages of scripts/prof_frame_reframed.py (a 1 M-point map handed over N times, nothing else running)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_rf
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o rf --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame_reframed.py ${1:-1000000} ${2:-12} > $O/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["TotalDurationNs"]) > 20000:
        print(r["Name"][:100].ljust(100), r["Calls"].rjust(5), "%9.1f us avg" % (float(r["AverageNs"]) / 1e3), " min %8.1f" % (float(r["MinNs"]) / 1e3))
PY
rm -rf $O
