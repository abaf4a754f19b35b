#!/bin/bash
# Developer aid: rocprofv3 kernel stats of the c3 (or, with the argument c5, c5) stage script.  Run on the GPU box from the repo root.
O=$GRAFT_REPO_ROOT/gpurun_out/${2:-c3prof}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/exp_c3_stages.py $1 > $O/log.txt 2>&1
cd $GRAFT_REPO_ROOT; tail -1 $O/log.txt
python3 - $O <<PY
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:30]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
