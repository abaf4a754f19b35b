#!/bin/bash
# Developer aid (GPU box): the scan's block radius (build flag RGC_SCAN_R) against the dependent c-main sequence.
cd "$GRAFT_REPO_ROOT"
for r in ${1:-1 2}; do
  RGC_EXTRA_FLAGS="-DRGC_SCAN_R=$r" python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
  for rep in 1 2; do
    python3 bench.py --configs none --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(json.dumps({'scan_r': $r, 'value': d['value'], 'one_frame': d['one_frame_at_a_time']['scans_per_s'], 'knn_in_frame_ms': d['roofline']['avg_launch_ms'], 'stages': d['kernel_ms_per_step']}))"
  done
done | tee gpurun_out/exp_scan_r.jsonl
python3 rgc-slam_amd/build.py --force > /dev/null 2>&1
