#!/bin/bash
# A/B of prebuilt libraries (exp_flags/librgc_<name>.so, "cur" = the product) on the bench's keys, same box, alternating:
#   bash scripts/ab_bench.sh <rounds> <name> ...
cd $GRAFT_REPO_ROOT
R=$1; shift
for r in $(seq 1 $R); do
for name in "$@"; do
if [ $name = cur ]; then unset RGC_HIP_LIB; else export RGC_HIP_LIB=$GRAFT_REPO_ROOT/exp_flags/librgc_$name.so; fi
timeout 600 python bench.py --steps 20 --warmup 4 --configs none --no-cpu-baseline --no-two-sequences 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', 'value', d['value'], 'alone', d['roofline']['launch_alone_ms'], 'in-run', d['roofline']['avg_launch_ms'], 'one-frame', d['one_frame_at_a_time']['ms_per_step'], 'steady', d['steady_state']['two_contexts']['ms_per_step'], d['steady_state']['one_frame_at_a_time']['ms_per_step'], 'lazy', d['lazy_target']['two_contexts']['scans_per_s'])"
done
done
