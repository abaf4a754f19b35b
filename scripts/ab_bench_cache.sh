#!/bin/bash
# A/B of the neighbour-list cache (RGC_KNN_CACHE=0/1 in the environment, one library) on the bench's keys, same box, alternating:
#   bash scripts/ab_bench_cache.sh <rounds>
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $1); do
for v in 0 1; do
RGC_KNN_CACHE=$v timeout 600 python bench.py --steps 20 --warmup 4 --configs none --no-cpu-baseline --no-two-sequences 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('cache=$v', 'value', d['value'], 'alone', d['roofline']['launch_alone_ms'], 'in-run', d['roofline']['avg_launch_ms'], 'one-frame', d['one_frame_at_a_time']['ms_per_step'], 'steady', d['steady_state']['two_contexts']['ms_per_step'], d['steady_state']['one_frame_at_a_time']['ms_per_step'], 'lazy', d['lazy_target']['two_contexts']['scans_per_s'])"
done
done
