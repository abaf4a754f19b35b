#!/bin/bash
# rocprofv3 kernel stats of the front-end alone (scripts/bench_frontend.py): average duration of every k_fe_* kernel
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fe_prof
rocprofv3 --kernel-trace --stats -d /tmp/fe_prof -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_frontend.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/fe_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_fe_" in r["Name"] or "scan" in r["Name"]:
        print(r["Name"].split("(")[0].replace("rgck::", "")[:28].ljust(28), r["Calls"].rjust(5), f'{float(r["AverageNs"]) / 1e3:8.1f} us')
PY
