#!/bin/bash
# Bulk kNN launch of the map with the y-slowest cell order (-DRGC_Y_SLOWEST=1) for several XCD run lengths (-DRGC_XCD_RUN): launch time and FETCH_SIZE (KB as rocprofv3 reports it).
cd "$GRAFT_REPO_ROOT"
for run in "$@"; do
  RGC_EXTRA_FLAGS="-DRGC_Y_SLOWEST=1 -DRGC_XCD_RUN=$run" python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
  t=$(python3 scripts/lab_knn.py 1000000 5 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['target']['knn_cov_target'])")
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/xr
  rocprofv3 --pmc FETCH_SIZE -d /tmp/xr -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py 1000000 3 > /dev/null 2>&1
  cd "$GRAFT_REPO_ROOT"
  python3 - <<PY
import csv, glob
v = [float(r["Counter_Value"]) for f in glob.glob("/tmp/xr/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "k_knn_sp<20, true>" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("XCD_RUN $run  launch ms $t  FETCH_SIZE KB", round(sum(v) / max(len(v), 1), 1), " => HBM MB (2 x fetch + 24 written)", round((2 * sum(v) / max(len(v), 1) * 1024 + 24.0e6) / 1e6, 1))
PY
done
python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
