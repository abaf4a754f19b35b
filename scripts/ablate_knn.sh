#!/bin/bash
# Where the bulk kNN kernel's instructions go: rebuilds the library with -DRGC_ABLATE=n (1: no chain inserts, 2: stop after the scan,
# 3: row table only), times the launch and counts its VALU instructions.  Results are wrong by construction: developer builds only.
# Run on the GPU box:  scripts/ablate_knn.sh   -> gpurun_out/ablate_knn.jsonl ; rebuild the product afterwards (python rgc-slam_amd/build.py --force)
cd "$GRAFT_REPO_ROOT"
: > gpurun_out/ablate_knn.jsonl
for n in 0 1 2 3; do
  RGC_EXTRA_FLAGS="-DRGC_ABLATE=$n" python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
  t=$(python3 scripts/lab_knn.py 1000000 5 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['target']['knn_cov_target'])")
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/abl_$n
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU -d /tmp/abl_$n -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/prof_frame.py 1000000 3 > /dev/null 2>&1
  cd "$GRAFT_REPO_ROOT"
  python3 - <<PY >> gpurun_out/ablate_knn.jsonl
import csv, glob, json, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/abl_$n/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_knn_sp<20, true>" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({"ablate": $n, "knn_cov_target_ms": $t, **{k: sum(v) / len(v) / 1e6 for k, v in acc.items()}}))
PY
done
RGC_EXTRA_FLAGS="" python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
cat gpurun_out/ablate_knn.jsonl
