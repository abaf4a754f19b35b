#!/bin/bash
# What the node's slow frames consist of: rocprofv3 kernel trace + HIP API trace of tests/cpp/test_odometry_node.cpp (mode: resident chain),
# the longest kernels and API calls with their start times, and the frames' wall times.   bash scripts/prof_node_frames.sh <resident 0|1> <chain 0|1>
cd "$GRAFT_REPO_ROOT"
[ -x /tmp/node ] || bash scripts/node_frames.sh 0 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pn
rocprofv3 --kernel-trace --hip-runtime-trace -d /tmp/pn -o run --output-format csv -- /tmp/node /tmp/sweeps.bin $1 1 50 $2 0 > /tmp/node.out 2>&1
grep "^pose" /tmp/node.out | awk '{printf "%s ", $NF}'; echo
python3 - <<'PY'
import csv, glob
k = glob.glob("/tmp/pn/**/*kernel_trace.csv", recursive=True)[0]
a = glob.glob("/tmp/pn/**/*hip_api_trace.csv", recursive=True)[0]
kr = list(csv.DictReader(open(k))); ar = list(csv.DictReader(open(a)))
t0 = min(int(r["Start_Timestamp"]) for r in ar)
print("longest kernels:")
for d, t, n in sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), (int(r["Start_Timestamp"]) - t0) / 1e6, r["Kernel_Name"][:70]) for r in kr), reverse=True)[:10]:
    print("  %9.3f ms at %10.3f ms  %s" % (d / 1e6, t, n))
print("longest HIP calls:")
for d, t, n in sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), (int(r["Start_Timestamp"]) - t0) / 1e6, r["Function"]) for r in ar), reverse=True)[:14]:
    print("  %9.3f ms at %10.3f ms  %s" % (d / 1e6, t, n))
PY
