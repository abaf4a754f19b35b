#!/bin/bash
# Developer aid (GPU box): rocprofv3 kernel statistics of scripts/bench_rolling.py (the keyframe store's commit: leaf filter + target rebuild).
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pr
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pr -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_rolling.py > /dev/null 2>&1
f=$(find /tmp/pr -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:30]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "%9.1f us avg" % (float(r["AverageNs"]) / 1e3), "%6.2f %%" % float(r["Percentage"]))
PY
