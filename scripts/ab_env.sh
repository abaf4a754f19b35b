#!/bin/bash
# A/B of environment knobs of one library on the bench's keys, same box, alternating:
#   bash scripts/ab_env.sh <rounds> "A=1 B=2" "A=0" ...       (each argument: the assignments of one variant, "-" = none)
cd $GRAFT_REPO_ROOT
R=$1; shift
for r in $(seq 1 $R); do
for v in "$@"; do
if [ "$v" = "-" ]; then e=""; else e="$v"; fi
env $e timeout 600 python bench.py --steps 20 --warmup 4 --configs none --no-cpu-baseline --no-two-sequences 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[$v]', 'value', d['value'], 'alone', d['roofline']['launch_alone_ms'], 'in-run', d['roofline']['avg_launch_ms'], 'one-frame', d['one_frame_at_a_time']['ms_per_step'], 'steady', d['steady_state']['two_contexts']['ms_per_step'], d['steady_state']['one_frame_at_a_time']['ms_per_step'], 'lazy', d['lazy_target']['two_contexts']['scans_per_s'])"
done
done
