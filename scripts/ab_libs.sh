#!/bin/bash
# A/B of two library builds on one box, alternating: bench.py's headline keys and the per-kernel breakdown.   usage: bash scripts/ab_libs.sh <libA.so|-> <libB.so|-> [rounds]   (- = the in-tree build)
cd "$(dirname "$0")/.." || exit 1
R=${3:-2}
for r in $(seq 1 $R); do
  for l in "$1" "$2"; do
    if [ "$l" = "-" ]; then unset RGC_HIP_LIB; else export RGC_HIP_LIB=$PWD/$l; fi
    python3 bench.py --configs none --no-cpu-baseline --no-two-sequences 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$l', 'value', d['value'], 'steady', d['steady_state']['two_contexts']['scans_per_s'], 'one_at_a_time', d['steady_state']['one_frame_at_a_time']['scans_per_s'], 'lazy', d['lazy_target']['two_contexts']['scans_per_s'], 'src_bulk', k['knn_cov_source'], 'src_coop', k['knn_coop_source'], 'checksum', d['final_pose_checksum'])"
  done
done
